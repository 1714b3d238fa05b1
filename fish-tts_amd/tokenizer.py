"""The reference tokenizer (fish_tts/models/tokenizer.py): token-id layout (25-101), the tiktoken-format rank file
loader (103-113), encode/decode (118-150) and from_pretrained (155-166) -- SURVEY.md §8-f F3.

The reference delegates the byte-pair encoding to `tiktoken` 0.12.0 (uv.lock:1127-1128), a third-party Rust wheel
that is NOT in this image; `BPETokenizer` restates its published algorithm in pure Python (regex pre-split with the
reference's FISH_TIKTOKEN_PATTERN, then greedy lowest-rank pair merging per piece, special tokens matched first).
There is no tokenizer.tiktoken file and no tiktoken here to compare against, so its merges are "parity unpinned"
(tests/test_host_logic.py pins the algorithm on hand-made rank tables).  `ByteTokenizer` is the synthetic stand-in
over the same id layout for runs without a rank file."""
from __future__ import annotations

import base64
import json
import re
from pathlib import Path

IM_END_TOKEN = "<|im_end|>"
MODALITY_TOKENS = {"text": "<|text|>", "voice": "<|voice|>", "interleave": "<|interleave|>"}
NAMED_SPECIAL_TOKENS = [
    "<|begin_of_text|>", "<|end_of_text|>", "<|pad|>", "<|im_start|>", "<|im_end|>", "<|phoneme_start|>",
    "<|phoneme_end|>", "<|tool_call_start|>", "<|tool_call_end|>", "<|text|>", "<|voice|>", "<|interleave|>",
    "<|audio_start|>", "<|audio_end|>", "<|audio|>"]
NUM_SEMANTIC = 4096
ALL_SPECIAL_TOKENS = NAMED_SPECIAL_TOKENS + [f"<|semantic:{i}|>" for i in range(NUM_SEMANTIC)]
S1_MINI_BPE_RANKS = 151643  # tests/test_config.py:77-110 of the reference pin BOS = 151643


class TokenLayout:
    """ids of the special tokens: BPE ranks first, then the specials in order (tokenizer.py:83-101)."""

    def __init__(self, n_ranks: int, special_tokens=None):
        special_tokens = list(special_tokens) if special_tokens is not None else ALL_SPECIAL_TOKENS
        self.n_ranks = n_ranks
        self.all_special_tokens_with_ids = {t: n_ranks + i for i, t in enumerate(special_tokens)}
        self.semantic_id_to_token_id = {}
        for t, i in self.all_special_tokens_with_ids.items():
            m = re.match(r"<\|semantic:(\d+)\|>", t)
            if m:
                self.semantic_id_to_token_id[int(m.group(1))] = i
        self.semantic_begin_id = self.semantic_id_to_token_id[0]
        self.semantic_end_id = self.semantic_id_to_token_id[max(self.semantic_id_to_token_id)]

    def get_token_id(self, token: str) -> int:
        return self.all_special_tokens_with_ids[token]


class ByteTokenizer(TokenLayout):
    """Synthetic tokenizer: special tokens are recognised, everything else maps byte -> id (mod n_ranks).
    Same surface as the reference's FishTokenizer where the hot path touches it
    (encode / get_token_id / semantic_* ; inference.py:182,546,555,632)."""

    def __init__(self, n_ranks: int = S1_MINI_BPE_RANKS, special_tokens=None):
        super().__init__(n_ranks, special_tokens)
        named = [t for t in self.all_special_tokens_with_ids if not t.startswith("<|semantic:")]
        self._split = re.compile("(" + "|".join(re.escape(t) for t in named) + r"|<\|semantic:\d+\|>)")

    def encode(self, s: str, allowed_special=True) -> list:
        out = []
        for piece in self._split.split(s):
            if not piece:
                continue
            if piece in self.all_special_tokens_with_ids:
                out.append(self.all_special_tokens_with_ids[piece])
            else:
                out.extend(b % self.n_ranks for b in piece.encode("utf-8"))
        return out


def load_tokenizer(model_dir):
    """tokenizer.tiktoken + special_tokens.json as FishTokenizer.from_pretrained (tokenizer.py:155-166);
    falls back to the byte tokenizer over the same id layout when tiktoken is not installed."""
    path = Path(model_dir)
    special = None
    if (path / "special_tokens.json").exists():
        with open(path / "special_tokens.json") as f:
            special = json.load(f)
    bpe = path / "tokenizer.tiktoken"
    if not bpe.exists():
        return ByteTokenizer(S1_MINI_BPE_RANKS, special)
    ranks = {}
    for line in open(bpe).read().splitlines():
        if not line:
            continue
        tok, rank = line.split()
        if tok == "=":
            continue
        ranks[base64.b64decode(tok)] = int(rank)
    try:
        import tiktoken  # noqa: F401
    except ImportError:
        return BPETokenizer(ranks, special)
    return _TiktokenTokenizer(path.name, ranks, special)


FISH_TIKTOKEN_PATTERN = "|".join([r"(?i:'s|'t|'re|'ve|'m|'ll|'d)", r"\p{P}", r"[^\r\n\p{L}\p{N}]?\p{L}+", r"\p{N}",
                                  r" ?[^\s\p{L}\p{N}]+[\r\n]*", r"\s*[\r\n]+", r"\s+(\?!\S)", r"\s+"])
TIKTOKEN_MAX_ENCODE_CHARS = 400_000


def load_tiktoken_bpe(path) -> dict:
    """tokenizer.py:103-113: lines of `<base64 token> <rank>`; a literal "=" token line is skipped."""
    ranks = {}
    for line in open(path).read().splitlines():
        if not line:
            continue
        tok, rank = line.split()
        if tok == "=":
            continue
        ranks[base64.b64decode(tok)] = int(rank)
    return ranks


class BPETokenizer(TokenLayout):
    """FishTokenizer without tiktoken: same constructor data (mergeable ranks + special-token list), same
    encode/decode surface (tokenizer.py:118-150)."""

    def __init__(self, ranks: dict, special_tokens=None, pattern: str = FISH_TIKTOKEN_PATTERN):
        import regex
        super().__init__(len(ranks), special_tokens)
        self._ranks = ranks
        self._pat = regex.compile(pattern)
        self._special_re = regex.compile("|".join(regex.escape(t) for t in self.all_special_tokens_with_ids))
        self._id_to_bytes = {i: b for b, i in ranks.items()}
        self._id_to_special = {i: t for t, i in self.all_special_tokens_with_ids.items()}
        self._cache = {}

    @property
    def vocab_size(self) -> int:
        return len(self._ranks)

    @property
    def num_special_tokens(self) -> int:
        return len(self.all_special_tokens_with_ids)

    @property
    def special_tokens_set(self) -> set:
        return set(self.all_special_tokens_with_ids)

    def _bpe(self, piece: bytes) -> list:
        hit = self._cache.get(piece)
        if hit is not None:
            return hit
        ranks = self._ranks
        if piece in ranks:
            out = [ranks[piece]]
        else:
            parts = [piece[i:i + 1] for i in range(len(piece))]
            while len(parts) > 1:
                best, best_rank = -1, None
                for i in range(len(parts) - 1):
                    r = ranks.get(parts[i] + parts[i + 1])
                    if r is not None and (best_rank is None or r < best_rank):
                        best, best_rank = i, r
                if best < 0:
                    break
                parts[best:best + 2] = [parts[best] + parts[best + 1]]
            out = [ranks[q] for q in parts]  # KeyError = a byte the rank file does not cover
        if len(self._cache) < 1 << 16:
            self._cache[piece] = out
        return out

    def encode(self, s: str, allowed_special=True) -> list:
        assert isinstance(s, str)
        if allowed_special is True:
            allowed = self.special_tokens_set
        elif allowed_special is False:
            allowed = set()
        else:
            allowed = set(allowed_special)
        out: list = []
        for c0 in range(0, len(s), TIKTOKEN_MAX_ENCODE_CHARS):
            sub = s[c0:c0 + TIKTOKEN_MAX_ENCODE_CHARS]
            start = 0
            while True:
                # next special token that is allowed; others are plain text (disallowed_special=set())
                m, scan = None, start
                while allowed:
                    m = self._special_re.search(sub, scan)
                    if m is None or m.group(0) in allowed:
                        break
                    scan = m.start() + 1
                    m = None
                end = m.start() if m is not None else len(sub)
                self._encode_ordinary_text(sub[start:end], out)
                if m is None:
                    break
                out.append(self.all_special_tokens_with_ids[m.group(0)])
                start = m.end()
        return out

    def _encode_ordinary_text(self, text: str, out: list) -> None:
        for m in self._pat.finditer(text):
            out.extend(self._bpe(m.group(0).encode("utf-8")))

    def decode(self, tokens) -> str:
        buf = bytearray()
        for t in tokens:
            t = int(t)
            if t in self._id_to_bytes:
                buf += self._id_to_bytes[t]
            else:
                buf += self._id_to_special[t].encode("utf-8")
        return buf.decode("utf-8", errors="replace")

    @classmethod
    def from_pretrained(cls, path) -> "BPETokenizer":
        path = Path(path)
        special = None
        if (path / "special_tokens.json").exists():
            with open(path / "special_tokens.json") as f:
                special = json.load(f)
        return cls(load_tiktoken_bpe(path / "tokenizer.tiktoken"), special)


class _TiktokenTokenizer(TokenLayout):
    PATTERN = "|".join([r"(?i:'s|'t|'re|'ve|'m|'ll|'d)", r"\p{P}", r"[^\r\n\p{L}\p{N}]?\p{L}+", r"\p{N}",
                        r" ?[^\s\p{L}\p{N}]+[\r\n]*", r"\s*[\r\n]+", r"\s+(\?!\S)", r"\s+"])

    def __init__(self, name, ranks, special):
        import tiktoken
        super().__init__(len(ranks), special)
        self._enc = tiktoken.core.Encoding(name=name, pat_str=self.PATTERN, mergeable_ranks=ranks,
                                           special_tokens=self.all_special_tokens_with_ids)

    def encode(self, s: str, allowed_special=True) -> list:
        allowed = self._enc.special_tokens_set if allowed_special is True else (allowed_special or set())
        out = []
        for i in range(0, len(s), 400_000):
            out += self._enc.encode(s[i:i + 400_000], allowed_special=allowed, disallowed_special=set())
        return out
