"""CPU-side host logic: prompt assembly against the reference's golden matrices, WAV/PCM packing,
VoiceProfile round trip, the public API surface, the singleton, utterance dealing."""
import inspect
import io
import os
import wave

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def test_prompt_matrices_match_reference():
    from fish_tts_amd.prompt import build_prompt
    from fish_tts_amd.tokenizer import ByteTokenizer
    gold = np.load(os.path.join(G, "prompt.npz"))
    tok = ByteTokenizer(151643)
    for name, nref in (("noref", 0), ("oneref", 1), ("tworef", 2)):
        texts = [bytes(gold[f"{name}.ref{i}.text"]).decode() for i in range(nref)]
        codes = [gold[f"{name}.ref{i}.codes"] for i in range(nref)]
        text = bytes(gold[f"{name}.text"]).decode()
        got = build_prompt(tok, text, texts, codes, 10)
        assert got.dtype == np.int32
        assert np.array_equal(got, gold[f"{name}.prompt"]), name


def test_token_layout_pins():
    """ids pinned by the reference's tests/test_config.py:77-110."""
    from fish_tts_amd.tokenizer import ByteTokenizer
    tok = ByteTokenizer()
    assert tok.get_token_id("<|begin_of_text|>") == 151643
    assert tok.get_token_id("<|audio_end|>") == 151656
    assert tok.semantic_begin_id == 151658 and tok.semantic_end_id == 155753
    assert tok.encode("<|im_end|>ab") == [151647, 97, 98]


def test_wav_and_pcm_packing():
    from fish_tts_amd.synthesizer import FishTTS
    audio = np.array([0.0, 0.5, -0.5, 1.5, -2.0, 0.999], dtype=np.float32)
    data = FishTTS._to_wav_bytes(audio)
    assert len(data) == 44 + 2 * len(audio)
    with wave.open(io.BytesIO(data), "rb") as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate()) == (1, 2, 44100)
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
    assert pcm.tolist() == [0, 16383, -16383, 32767, -32767, 32734]  # clipped first (synthesizer.py:638)


def test_voice_profile_roundtrip(tmp_path):
    from fish_tts_amd import VoiceProfile
    codes = np.random.default_rng(0).integers(0, 1024, size=(10, 50)).astype(np.int64)
    p = VoiceProfile(codes=codes, text="hello", name="v")
    f = tmp_path / "voice.npy"
    p.save(f)
    q = VoiceProfile.load(f, text="hello")
    assert q.name == "voice" and q.text == "hello" and q.codes.dtype == np.int64
    assert np.array_equal(q.codes, codes)


def test_public_api_surface():
    import fish_tts_amd as ft
    sig = inspect.signature(ft.get_instance)
    assert list(sig.parameters) == ["model_dir", "device", "precision", "warmup"]
    assert [p.default for p in sig.parameters.values()] == [None, "cuda", "bf16", True]
    sig = inspect.signature(ft.FishTTS.synthesize)
    assert list(sig.parameters) == ["self", "text", "references", "temperature", "top_p", "repetition_penalty", "max_tokens"]
    assert [p.default for p in list(sig.parameters.values())[2:]] == [None, 0.7, 0.8, 1.1, 2048]
    sig = inspect.signature(ft.FishTTS.synthesize_stream)
    assert list(sig.parameters) == ["self", "text", "references", "chunk_tokens", "min_first_chunk", "kwargs"]
    assert sig.parameters["chunk_tokens"].default == 20 and sig.parameters["min_first_chunk"].default == 10
    for name in ("set_references", "add_reference", "clear_references", "get_references", "num_references",
                 "encode_reference", "sample_rate", "precision"):
        assert hasattr(ft.FishTTS, name), name


def test_singleton_ignores_later_arguments(monkeypatch):
    import fish_tts_amd.synthesizer as S
    made = []

    class Fake:
        def __init__(self, **kw):
            made.append(kw)
    monkeypatch.setattr(S, "FishTTS", Fake)
    S.reset_instance()
    a = S.get_instance(model_dir="x", warmup=False)
    b = S.get_instance(model_dir="y", precision="fp32")
    assert a is b and len(made) == 1 and made[0]["model_dir"] == "x"
    S.reset_instance()
    c = S.get_instance(model_dir="z")
    assert c is not a and len(made) == 2
    S.reset_instance()


def test_cpu_device_and_missing_model_dir_fail_loudly():
    from fish_tts_amd import FishTTS
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FishTTS(model_dir="whatever", device="cpu")
    with pytest.raises(RuntimeError, match="model_dir is required"):
        FishTTS(model_dir=None, device="cuda", warmup=False)


def test_deal_utterances_balances_lengths():
    from fish_tts_amd.parallel import deal_utterances
    lens = [430, 108, 300, 250, 120, 400, 200, 180]
    parts = deal_utterances(lens, 4)
    assert sorted(i for p in parts for i in p) == list(range(8))
    loads = [sum(lens[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lens) // 2


def test_config_defaults_follow_reference_rules():
    from fish_tts_amd.config import DualARModelArgs, s1_mini_args
    a = DualARModelArgs(dim=512, n_head=8, head_dim=None, intermediate_size=None)
    assert a.head_dim == 64 and a.n_local_heads == 8 and a.intermediate_size == 1536 and a.fast_dim == 512
    b = s1_mini_args()
    assert (b.dim, b.n_layer, b.n_head, b.num_codebooks, b.codebook_size, b.vocab_size) == (1024, 28, 16, 10, 4096, 155776)


def test_prompt_prefix_split_is_text_independent():
    """build_prompt_split: the leading n_prefix columns are [interleave, (speaker, ref text, ref codes, im_end)*]
    and do not change with the text to speak (what the reference-prefix K/V cache keys on)."""
    from fish_tts_amd.prompt import build_prompt, build_prompt_split
    from fish_tts_amd.tokenizer import ByteTokenizer
    tok = ByteTokenizer()
    rng = np.random.default_rng(0)
    codes = np.concatenate([rng.integers(0, 4096, (1, 37)), rng.integers(0, 1024, (9, 37))]).astype(np.int32)
    a, na = build_prompt_split(tok, "Hello world", ["ref transcript"], [codes], 10)
    b, nb = build_prompt_split(tok, "Something else entirely.", ["ref transcript"], [codes], 10)
    assert na == nb and na > 37
    assert np.array_equal(a[:, :na], b[:, :nb])
    assert a[0, na - 1] == tok.get_token_id("<|im_end|>")
    assert np.array_equal(a, build_prompt(tok, "Hello world", ["ref transcript"], [codes], 10))
    c, nc = build_prompt_split(tok, "Hello world", [], [], 10)
    assert nc == 0 and c.shape[1] < a.shape[1]
    d, nd = build_prompt_split(tok, "Hello world", None, None, 10)
    assert nd == 0 and np.array_equal(c, d)


def _toy_ranks():
    ranks = {bytes([i]): i for i in range(256)}
    for i, t in enumerate([b"he", b"ll", b"hell", b"hello", b" w", b"or", b" wor", b"ld", b"ab", b"bc"]):
        ranks[t] = 256 + i
    return ranks


def test_bpe_tokenizer_algorithm():
    """the published tiktoken algorithm on a hand-made rank table (tiktoken itself is absent:
    parity with it is unpinned).  Regex pre-split, lowest-rank pair first (leftmost on ties), whole-piece lookup,
    special tokens matched before text, disallowed specials encoded as text, decode round trip."""
    from fish_tts_amd.tokenizer import BPETokenizer
    tok = BPETokenizer(_toy_ranks(), ["<|a|>", "<|semantic:0|>", "<|semantic:1|>"])
    assert tok.vocab_size == 266 and tok.num_special_tokens == 3
    assert tok.get_token_id("<|a|>") == 266 and tok.semantic_begin_id == 267 and tok.semantic_end_id == 268
    assert tok.encode("hello") == [259]                              # whole piece is a token
    assert tok.encode("hell") == [258] and tok.encode("hel") == [256, ord("l")]
    assert tok.encode(" world") == [262, 263]                        # " wor" + "ld": merges by rank, not left to right
    assert tok.encode("abc") == [264, ord("c")]                      # "ab" (264) outranks "bc" (265)
    assert tok.encode("don't") == [ord("d"), ord("o"), ord("n"), ord("'"), ord("t")]   # contraction is its own piece
    ids = tok.encode("hello world<|a|>x<|semantic:1|>")
    assert ids == [259, 262, 263, 266, ord("x"), 268]
    assert tok.decode(ids) == "hello world<|a|>x<|semantic:1|>"
    assert tok.encode("<|a|>x", allowed_special=False) == [ord(c) for c in "<|a|>x"]
    assert tok.encode("<|a|><|semantic:0|>", allowed_special={"<|semantic:0|>"}) == [ord(c) for c in "<|a|>"] + [267]
    text = "Grüße, 世界! 123\n\n  tabs\tand  spaces "
    assert tok.decode(tok.encode(text)) == text                       # byte-level: lossless on any UTF-8
    assert tok.encode("") == []


def test_bpe_tokenizer_against_an_independent_bpe_engine():
    """F3 beyond self-made cases: `tiktoken` and the model's rank file are absent, so parity with them stays unpinned -
    but the algorithm can be held against an INDEPENDENT engine.  A byte-level BPE is trained with the `tokenizers`
    package (Rust merge loop, Oniguruma regex) behind the reference's FISH_TIKTOKEN_PATTERN pre-split
    (tokenizer.py:16-27); its vocabulary, turned into a tiktoken-style rank table (token bytes -> id, merges ranked in
    training order), is given to BPETokenizer: both must encode 2 500 texts (corpus lines and random strings with
    contractions, digits, CJK, umlauts, tabs and newlines) to the same ids."""
    import random
    tk = pytest.importorskip("tokenizers")
    from tokenizers import Regex, Tokenizer, models, pre_tokenizers, trainers
    from fish_tts_amd.tokenizer import FISH_TIKTOKEN_PATTERN, BPETokenizer
    rnd = random.Random(0)
    words = ["hello", "world", "the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "Grüße", "世界", "naïve", "don't",
             "it's", "we'll", "123", "4567", "foo_bar", "x=y+z;", "\n\n", "  tabs\t", "über", "señor", "日本語", "テキスト", "emoji😀",
             "they've", "I'm", "you'd"]
    corpus = [" ".join(rnd.choice(words) for _ in range(rnd.randint(3, 12))) + rnd.choice(["", ".", "!", "?\n", " \n\n", "  "])
              for _ in range(3000)]
    hf = Tokenizer(models.BPE())
    hf.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(FISH_TIKTOKEN_PATTERN), behavior="isolated"),
                                                pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    hf.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=256 + 400, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                                       special_tokens=[], show_progress=False))
    # the byte <-> printable-character table of byte-level BPE vocabularies (GPT-2's bytes_to_unicode)
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, extra = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + extra)
            extra += 1
    to_byte = {chr(c): b for b, c in zip(bs, cs)}
    vocab = hf.get_vocab()
    ranks = {bytes(to_byte[ch] for ch in t): i for t, i in vocab.items()}
    assert len(ranks) == len(vocab) > 300
    mine = BPETokenizer(ranks, ["<|semantic:0|>"])
    texts = corpus[:500] + ["".join(rnd.choice("abc xyz\n\t'.,!?123üß世") for _ in range(rnd.randint(0, 40))) for _ in range(2000)]
    for t in texts:
        assert mine.encode(t, allowed_special=False) == hf.encode(t).ids, repr(t)


def test_tiktoken_file_loader_and_from_pretrained(tmp_path):
    import base64
    import json
    from fish_tts_amd.tokenizer import BPETokenizer, load_tiktoken_bpe, load_tokenizer
    ranks = _toy_ranks()
    lines = [f"{base64.b64encode(t).decode()} {r}" for t, r in ranks.items()]
    lines.insert(3, "")            # blank lines and a literal "=" token line are skipped (tokenizer.py:106-111)
    lines.insert(7, "= 999")
    (tmp_path / "tokenizer.tiktoken").write_text("\n".join(lines))
    assert load_tiktoken_bpe(tmp_path / "tokenizer.tiktoken") == ranks
    (tmp_path / "special_tokens.json").write_text(json.dumps(["<|im_end|>", "<|semantic:0|>", "<|semantic:1|>"]))
    tok = BPETokenizer.from_pretrained(tmp_path)
    assert tok.get_token_id("<|im_end|>") == 266 and tok.encode("hello<|im_end|>") == [259, 266]
    try:
        import tiktoken  # noqa: F401
    except ImportError:
        assert isinstance(load_tokenizer(tmp_path), BPETokenizer)
    (tmp_path / "special_tokens.json").unlink()
    full = BPETokenizer.from_pretrained(tmp_path)                      # default = the reference's ALL_SPECIAL_TOKENS
    assert full.num_special_tokens == 15 + 4096 and full.get_token_id("<|begin_of_text|>") == 266
    assert full.semantic_end_id - full.semantic_begin_id == 4095


def test_checkpoint_key_layout_normalisation():
    """model.pth variants the reference accepts (llama.py:476-498, 222-227): a "state_dict" wrapper,
    the "model." prefix, audio_* tensors to drop, separate wq/wk/wv to fuse."""
    import torch
    from fish_tts_amd.ar_engine import normalise_state_dict
    g = torch.Generator().manual_seed(0)
    wq, wk, wv = (torch.randn(8, 4, generator=g), torch.randn(4, 4, generator=g), torch.randn(4, 4, generator=g))
    raw = {"state_dict": {"model.layers.0.attention.wq.weight": wq, "model.layers.0.attention.wk.weight": wk,
                          "model.layers.0.attention.wv.weight": wv, "model.layers.0.attention.wo.weight": torch.ones(4, 8),
                          "model.audio_projector.weight": torch.zeros(2, 2), "model.embeddings.weight": torch.zeros(3, 4)}}
    sd = normalise_state_dict(raw)
    assert set(sd) == {"layers.0.attention.wqkv.weight", "layers.0.attention.wo.weight", "embeddings.weight"}
    assert torch.equal(sd["layers.0.attention.wqkv.weight"], torch.cat([wq, wk, wv]))
    plain = {"layers.0.attention.wqkv.weight": torch.cat([wq, wk, wv]), "embeddings.weight": torch.zeros(3, 4)}
    assert set(normalise_state_dict(plain)) == set(plain)


def test_codec_weight_norm_folding_matches_torch():
    """codec.pth keeps weight-normed convs as parametrizations.weight.original0/1 under a "generator." prefix
    (vocoder.py:423-429, synthesizer.py:276-282); folded weights must equal what torch materialises."""
    import torch
    import torch.nn as nn
    from torch.nn.utils.parametrizations import weight_norm
    from fish_tts_amd.codec_engine import fold_weight_norm
    torch.manual_seed(0)
    conv = weight_norm(nn.Conv1d(6, 10, 3))
    convt = weight_norm(nn.ConvTranspose1d(6, 4, 4, stride=2))
    with torch.no_grad():
        conv.parametrizations.weight.original0.mul_(1.7)
        convt.parametrizations.weight.original0.add_(0.3)
    sd = {"generator.a." + k: v for k, v in conv.state_dict().items()}
    sd.update({"generator.b." + k: v for k, v in convt.state_dict().items()})
    sd["discriminator.junk"] = torch.zeros(1)
    sd["generator.c.alpha"] = torch.ones(1, 6, 1)
    out = fold_weight_norm({"state_dict": sd})
    assert set(out) == {"a.weight", "a.bias", "b.weight", "b.bias", "c.alpha"}
    assert torch.allclose(out["a.weight"], conv.weight.detach(), atol=1e-6)
    assert torch.allclose(out["b.weight"], convt.weight.detach(), atol=1e-6)


def test_stream_does_not_deadlock_when_ar_outruns_codec(monkeypatch):
    """synthesize_stream's two bounded queues (synthesizer.py:483-584): with an AR loop faster than the codec the
    reference's put()/join() order can deadlock (worker blocked on a full audio queue while the producer waits);
    here the producer keeps draining.  Simulated on the CPU: instant frames, a slow decoder, 14 chunks."""
    import threading
    import time
    import fish_tts_amd as ft
    import fish_tts_amd.generation as gen
    from fish_tts_amd.synthesizer import _PrefillCache

    def fake_generate_long(**kw):
        for i in range(131):
            yield gen.GenerateResponse(action="sample", codes=np.full((10, 1), i, dtype=np.int32), text=kw["text"])
        yield gen.GenerateResponse(action="next")

    synth = ft.FishTTS.__new__(ft.FishTTS)
    synth._engine = synth._tokenizer = object()
    synth._prefix_cache = None
    synth._prefill_cache, synth._prefill_lock, synth._gen_lock = _PrefillCache(), threading.Lock(), threading.Lock()

    def slow_decode(codes):
        time.sleep(0.02)
        return np.asarray(codes)[0].astype(np.int16).tobytes()   # 2 bytes per frame: the frame indices of the chunk
    synth._decode_to_pcm = slow_decode
    monkeypatch.setattr(gen, "generate_long", fake_generate_long)
    out = []
    t = threading.Thread(target=lambda: out.extend(synth.synthesize_stream("x", chunk_tokens=10, min_first_chunk=4)), daemon=True)
    t.start()
    t.join(20)
    assert not t.is_alive(), "synthesize_stream deadlocked"
    frames = np.frombuffer(b"".join(out), dtype=np.int16)
    assert np.array_equal(frames, np.arange(131))                 # every frame, in order
    assert [len(c) // 2 for c in out] == [4] + [10] * 12 + [7]


def test_stream_raises_and_frees_the_lock_when_the_decoder_worker_dies(monkeypatch):
    """The decoder thread's exception must stop generation and reach the caller (synthesizer.py:522-523, 583-584)
    instead of leaving the producer spinning on a queue nobody drains, and the generation lock must be free after."""
    import threading
    import fish_tts_amd as ft
    import fish_tts_amd.generation as gen
    from fish_tts_amd.synthesizer import _PrefillCache

    produced = []

    def fake_generate_long(**kw):
        for i in range(2000):
            produced.append(i)
            yield gen.GenerateResponse(action="sample", codes=np.full((10, 1), i, dtype=np.int32), text=kw["text"])
        yield gen.GenerateResponse(action="next")

    synth = ft.FishTTS.__new__(ft.FishTTS)
    synth._engine = synth._tokenizer = object()
    synth._prefix_cache = None
    synth._prefill_cache, synth._prefill_lock, synth._gen_lock = _PrefillCache(), threading.Lock(), threading.Lock()
    calls = []

    def failing_decode(codes):
        calls.append(1)
        if len(calls) == 2:
            raise RuntimeError("codec exploded")
        return np.asarray(codes)[0].astype(np.int16).tobytes()
    synth._decode_to_pcm = failing_decode
    monkeypatch.setattr(gen, "generate_long", fake_generate_long)
    out, err = [], []

    def run():
        try:
            out.extend(synth.synthesize_stream("x", chunk_tokens=10, min_first_chunk=4))
        except Exception as e:  # noqa: BLE001
            err.append(e)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(20)
    assert not t.is_alive(), "synthesize_stream hangs after the decoder worker died"
    assert len(err) == 1 and isinstance(err[0], RuntimeError) and "codec exploded" in str(err[0])
    assert len(produced) < 200                      # generation stopped soon after the failure
    assert synth._gen_lock.acquire(timeout=1)       # released
    synth._gen_lock.release()
    assert b"".join(out) == np.arange(4, dtype=np.int16).tobytes()   # the first chunk came through


class FakeEngine:
    """Stands in for ARHipEngine in the scheduler tests: frame (1000 uid + count) for every row, <|im_end|> at eos_at."""
    max_batch, R, im_end_id, max_new_tokens = 3, 11, 99, 64

    def __init__(self):
        self.slot_utt, self.count, self.widths, self.parked, self.prefills = {}, {}, [], [], []

    def _sampling(self, *a):
        return a

    def _clamp_new(self, T, n):
        return min(n, self.max_new_tokens)

    def _frame(self, slot):
        self.count[slot] += 1
        uid, eos_at = self.slot_utt[slot]
        f = np.full(self.R, 1000 * uid + self.count[slot], dtype=np.int32)
        if eos_at and self.count[slot] == eos_at:
            f[0] = self.im_end_id
        return f

    def _start(self, prompt, sp, prefix, slot):
        self.prefills.append((int(prompt[0, 0]), slot))
        self.slot_utt[slot], self.count[slot] = (int(prompt[0, 0]), int(prompt[0, 1])), 0
        return self._frame(slot)

    def prefill_many(self, prompts, sps, slots, prefixes):
        return np.stack([self._start(p, s, None, slot) for p, s, slot in zip(prompts, sps, slots)])

    def park(self, slot):
        self.parked.append(slot)
        self.slot_utt.pop(slot, None)

    def decode(self, k, sps, poll):
        self.widths.append(len(sps))
        frames = np.zeros((len(sps), k, self.R), dtype=np.int32)
        n = np.zeros(len(sps), dtype=np.int32)
        for s in range(len(sps)):
            if s not in self.slot_utt:
                continue
            for j in range(k):
                frames[s, j] = self._frame(s)
                n[s] = j + 1
                if frames[s, j, 0] == self.im_end_id:
                    self.slot_utt.pop(s)
                    break
        return frames, n


def _fake_utt(uid, budget, eos_at=0):
    from fish_tts_amd.batch import Utterance
    p = np.zeros((11, 4), dtype=np.int32)
    p[0, 0], p[0, 1] = uid, eos_at
    return Utterance(p, budget)


def test_batch_scheduler_on_a_fake_engine():
    """fish_tts_amd.batch.run_batch without a GPU: a fake engine that emits frame counters checks the host policy -
    longest budget first into the lowest slots, one lock-step pass for the initial first frames, refill of finished
    slots, idle slots parked, burst width = 1 + highest active slot, budgets and <|im_end|> respected."""
    from fish_tts_amd.batch import run_batch
    utt = _fake_utt
    eng = FakeEngine()
    utts = [utt(1, 5), utt(2, 30), utt(3, 12, eos_at=7), utt(4, 9), utt(5, 20)]
    seen = []
    run_batch(eng, utts, burst=4, on_frames=lambda i, blk: seen.append((i, blk.shape[1])))
    got = [u.columns().shape[1] for u in utts]
    assert got == [5, 30, 7, 9, 20]                                  # budgets, and utterance 3 stops at its <|im_end|>
    assert [u.columns()[1, -1] % 1000 for u in utts] == got           # frames arrive in order, none lost or repeated
    assert eng.prefills[:3] == [(2, 0), (5, 1), (3, 2)]               # longest budgets first, into the lowest slots
    assert {uid for uid, _ in eng.prefills[3:]} == {4, 1}             # the rest refill whatever frees up
    assert eng.widths[0] == 3 and eng.widths[-1] == 1                 # the lock-step width narrows while the queue drains
    assert sorted(set(eng.parked)) == [0, 1, 2] and sum(n for _, n in seen) == sum(got)   # every slot ends parked
    run_batch(eng, [], burst=4)                                       # nothing to do: every slot parked, no decode


def test_refill_slots_are_grouped_into_contiguous_runs():
    """ARHipEngine.prefill_many draws the first frames of a refill per contiguous run of slots (ft_ar_first_frames takes a
    slot range): the grouping, as indices into the caller's order."""
    from fish_tts_amd.ar_engine import contiguous_runs
    assert contiguous_runs([]) == []
    assert contiguous_runs([4]) == [[0]]
    assert contiguous_runs([0, 1, 2, 3]) == [[0, 1, 2, 3]]
    assert contiguous_runs([5, 2, 3, 9]) == [[1, 2], [0], [3]]
    assert contiguous_runs([7, 6, 5, 1, 0, 3]) == [[4, 3], [5], [2, 1, 0]]


def test_batch_streams_deal_and_collect_on_fake_engines():
    """fish_tts_amd.batch.run_batch_streams without a GPU: several lock-step batches side by side (one engine and one host
    thread each) - utterances dealt longest budget first to the least loaded engine, callbacks carry indices into the
    caller's list, every utterance gets its budget (or stops at its <|im_end|>), an engine-side error surfaces."""
    from fish_tts_amd.batch import run_batch_streams
    engs = [FakeEngine(), FakeEngine()]
    budgets = [5, 30, 12, 9, 20, 7, 7, 16, 3]
    utts = [_fake_utt(i + 1, b, eos_at=7 if i == 2 else 0) for i, b in enumerate(budgets)]
    frames_of, finished = {}, []
    stats = run_batch_streams(engs, utts, burst=4, on_frames=lambda i, blk: frames_of.__setitem__(i, frames_of.get(i, 0) + blk.shape[1]),
                              on_done=finished.append)
    want = [5, 30, 7, 9, 20, 7, 7, 16, 3]
    assert [u.columns().shape[1] for u in utts] == want
    assert [frames_of[i] for i in range(len(utts))] == want and sorted(finished) == list(range(len(utts)))
    for i, u in enumerate(utts):                     # the frames are the utterance's own, in order
        assert list(u.columns()[1] // 1000) == [i + 1] * want[i] and list(u.columns()[1] % 1000) == list(range(1, want[i] + 1))
    shares = [sorted({uid for uid, _ in e.prefills}) for e in engs]
    assert sorted(shares[0] + shares[1]) == list(range(1, 10)) and not set(shares[0]) & set(shares[1])
    loads = [sum(budgets[uid - 1] for uid in sh) for sh in shares]
    assert abs(loads[0] - loads[1]) <= max(budgets) // 2, loads     # longest first to the least loaded: 55 / 54 here
    assert len(stats) == 2 and all(st["frame_steps"] > 0 for st in stats)
    assert run_batch_streams([FakeEngine()], [], burst=4) == [{"frame_steps": 0, "slot_frames": 0}]

    class Broken(FakeEngine):
        def decode(self, k, sps, poll):
            raise RuntimeError("device lost")
    try:
        run_batch_streams([FakeEngine(), Broken()], [_fake_utt(1, 5), _fake_utt(2, 6)], burst=4)
        raise AssertionError("the engine's error must surface")
    except RuntimeError as e:
        assert "device lost" in str(e)


def test_prefix_cache_is_lru_and_keyed_by_content():
    """generation.PrefixCache without a GPU: one K/V build per distinct prefix, hits move to the back of the eviction
    order, the least recently used entry is freed beyond the capacity, freed handles are rebuilt."""
    from fish_tts_amd.generation import PrefixCache

    class Handle:
        def __init__(self, owner, cols):
            self.owner, self.handle, self.n_pos = owner, object(), cols.shape[1]

        def free(self):
            self.handle = None
            self.owner.freed += 1

    class FakeEngine:
        def __init__(self):
            self.built, self.freed = 0, 0

        def build_prefix(self, cols):
            self.built += 1
            return Handle(self, cols)
    eng = FakeEngine()
    cache = PrefixCache(capacity=2, min_positions=4)
    a, b, c = (np.full((11, 6), v, dtype=np.int32) for v in (1, 2, 3))
    ha = cache.get(eng, a)
    assert cache.get(eng, a.copy()) is ha and eng.built == 1            # same content, another array: a hit
    cache.get(eng, b)
    cache.get(eng, a)                                                   # a is now the most recent
    cache.get(eng, c)                                                   # evicts b
    assert eng.built == 3 and eng.freed == 1 and len(cache) == 2
    assert cache.get(eng, a) is ha and eng.built == 3
    cache.get(eng, b)                                                   # rebuilt, evicts c
    assert eng.built == 4 and eng.freed == 2
    ha.free()                                                           # freed behind the cache's back (engine closed)
    assert cache.get(eng, a) is not ha and eng.built == 5
    cache.clear()
    assert len(cache) == 0
