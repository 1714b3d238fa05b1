"""Token-id layout of the reference tokenizer (fish_tts/models/tokenizer.py:25-101) and a byte-level
stand-in for synthetic runs.  The BPE itself (tiktoken, a third-party Rust wheel) is outside the hot
path; when `tiktoken` and a `tokenizer.tiktoken` file are present `load_tokenizer` uses them."""
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
        import tiktoken
    except ImportError:
        return ByteTokenizer(len(ranks), special)
    return _TiktokenTokenizer(path.name, ranks, special)


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
