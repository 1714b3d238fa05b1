"""End-to-end on the GPU through the public API (FishTTS.synthesize / synthesize_stream / references)
with tiny synthetic models, checked against the oracle pipeline (AR oracle -> codec oracle)."""
import dataclasses
import io
import wave

import numpy as np
import pytest
import torch

from oracle import ar as OA
from oracle import codec as OC
from tests.hip_util import args_from_shape
from tests.shapes import tiny_shape
from tests.test_codec_gpu import args_from_shape as codec_args_from_shape

pytestmark = pytest.mark.gpu


def api_codec_shape():
    return OC.CodecShape(n_codebooks=9, codebook_size=1024, semantic_codebook_size=2048, codebook_dim=8, latent_dim=64,
                         n_tf_layer=2, tf_n_head=4, tf_head_dim=16, tf_ffn=96, tf_window=8, tf_block_size=256,
                         upsample=[2, 2], decoder_dim=128, rates=[4, 2])


@pytest.fixture(scope="module")
def tts():
    import fish_tts_amd as ft
    from fish_tts_amd.tokenizer import NAMED_SPECIAL_TOKENS, ByteTokenizer
    # generate_long refuses prompts longer than max_seq_len - 2048 (inference.py:794): give the tiny model room
    shape = dataclasses.replace(tiny_shape(), max_seq_len=2304)
    tok = ByteTokenizer(256, NAMED_SPECIAL_TOKENS + [f"<|semantic:{i}|>" for i in range(2048)])
    assert tok.semantic_begin_id == shape.semantic_begin_id and tok.get_token_id("<|im_end|>") == shape.im_end_id
    cshape = api_codec_shape()
    # engines get the oracle's seeded weights so that both sides run the same model
    synth = ft.FishTTS.__new__(ft.FishTTS)
    ft.FishTTS.__init__(synth, None, "cuda", "fp32", False,
                        _synthetic=dict(args=args_from_shape(shape), tokenizer=tok, codec_args=codec_args_from_shape(cshape),
                                        with_codec=False, max_new_tokens=96))
    w = OA.random_weights(shape, seed=0)
    synth._engine.close()
    from fish_tts_amd.ar_engine import ARHipEngine
    from fish_tts_amd.codec_engine import CodecHipEngine
    synth._engine = ARHipEngine(args_from_shape(shape), shape.semantic_begin_id, shape.semantic_end_id, shape.im_end_id,
                                precision="fp32", max_new_tokens=96)
    synth._engine.load_state_dict(w)
    cw = OC.random_weights(cshape, seed=0)
    synth._vocoder = CodecHipEngine(codec_args_from_shape(cshape), max_frames=96)
    synth._vocoder.load_state_dict(cw)
    orc = OA.AROracle(shape, w, torch.float32)
    corc = OC.CodecOracle(cshape, cw)
    yield synth, tok, orc, corc, shape, cshape
    synth._engine.close()
    synth._vocoder.close()


def _oracle_codes(orc, prompt, n, **kw):
    orc.reset()
    seq = orc.generate(torch.from_numpy(prompt), n, **kw).numpy()
    return seq


def test_synthesize_matches_oracle_pipeline(tts):
    synth, tok, orc, corc, shape, cshape = tts
    from fish_tts_amd.prompt import build_prompt
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    wav = synth.synthesize("Hi there", max_tokens=12, **kw)
    prompt = build_prompt(tok, "Hi there", [], [], 10)
    seq = _oracle_codes(orc, prompt, 12, **kw)
    codes = seq[1:, prompt.shape[1]:-1]  # batch mode drops the last column (inference.py:839)
    want, _ = corc.decode(torch.from_numpy(codes.astype(np.int64))[None], torch.tensor([codes.shape[1]]))
    want = want[0, 0].numpy()
    with wave.open(io.BytesIO(wav), "rb") as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate()) == (1, 2, 44100)
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16).astype(np.float32) / 32767
    assert pcm.shape == want.shape == (codes.shape[1] * cshape.frame_len,)
    err = np.sqrt(np.mean((pcm - np.clip(want, -1, 1)) ** 2)) / np.sqrt(np.mean(want ** 2))
    assert err <= 3e-2, err


def test_stream_yields_every_frame_in_reference_chunking(tts):
    synth, tok, orc, corc, shape, cshape = tts
    from fish_tts_amd.prompt import build_prompt
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    chunks = list(synth.synthesize_stream("Hi there", chunk_tokens=7, min_first_chunk=3, max_tokens=20, **kw))
    prompt = build_prompt(tok, "Hi there", [], [], 10)
    seq = _oracle_codes(orc, prompt, 20, **kw)
    n = seq.shape[1] - prompt.shape[1]  # streaming yields all n frames (inference.py:721, 271)
    fl = cshape.frame_len
    sizes = [len(c) // (2 * fl) for c in chunks]
    want_sizes = [3] + [7] * ((n - 3) // 7) + ([(n - 3) % 7] if (n - 3) % 7 else [])
    assert sizes == want_sizes, (sizes, want_sizes)
    # chunks are decoded independently from zero context: compare the first chunk to the oracle
    first = np.frombuffer(chunks[0], dtype=np.int16).astype(np.float32) / 32767
    codes0 = seq[1:, prompt.shape[1]: prompt.shape[1] + 3].astype(np.int64)
    want0, _ = corc.decode(torch.from_numpy(codes0)[None], torch.tensor([3]))
    err = np.sqrt(np.mean((first - want0[0, 0].numpy()) ** 2)) / np.sqrt(np.mean(want0[0, 0].numpy() ** 2))
    assert err <= 3e-2, err


def test_references_enter_the_prompt(tts):
    synth, tok, orc, corc, shape, cshape = tts
    import fish_tts_amd as ft
    from fish_tts_amd.generation import generate_long
    from fish_tts_amd.prompt import build_prompt
    g = np.random.default_rng(4)
    ref = np.zeros((10, 6), dtype=np.int64)
    ref[0] = g.integers(0, 2048, 6)
    ref[1:] = g.integers(0, 1024, (9, 6))
    prof = ft.VoiceProfile(codes=ref, text="ref text", name="r")
    synth.set_references([prof])
    assert synth.num_references == 1 and synth.get_references()[0].name == "r"
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    res = [r for r in generate_long(engine=synth._engine, tokenizer=tok, text="Yo", max_new_tokens=8,
                                    prompt_text=["ref text"], prompt_tokens=[ref], **kw) if r.action == "sample"]
    prompt = build_prompt(tok, "Yo", ["ref text"], [ref], 10)
    seq = _oracle_codes(orc, prompt, 8, **kw)
    assert np.array_equal(res[0].codes, seq[1:, prompt.shape[1]:-1])
    wav_with = synth.synthesize("Yo", max_tokens=8, **kw)
    synth.clear_references()
    wav_without = synth.synthesize("Yo", max_tokens=8, **kw)
    assert wav_with != wav_without
    with pytest.raises(AssertionError, match="top_p"):
        synth.synthesize("x", top_p=0.0)


def test_reference_kv_is_cached_per_voice_and_changes_nothing(tts):
    """SURVEY §8-f F1 through the public API: the K/V of the reference part of the prompt is computed once per voice;
    codes equal the oracle's on the full prompt (fp32: exact)."""
    synth, tok, orc, corc, shape, cshape = tts
    import fish_tts_amd as ft
    from fish_tts_amd.generation import generate_long
    from fish_tts_amd.prompt import build_prompt_split
    g = np.random.default_rng(9)
    ref = np.concatenate([g.integers(0, 2048, (1, 40)), g.integers(0, 1024, (9, 40))]).astype(np.int64)
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    cache = synth._prefix_cache
    cache.clear()
    for text in ("Yo", "Another sentence"):
        res = [r for r in generate_long(engine=synth._engine, tokenizer=tok, text=text, max_new_tokens=8,
                                        prompt_text=["ref text"], prompt_tokens=[ref], prefix_cache=cache, **kw)
               if r.action == "sample"]
        prompt, n_prefix = build_prompt_split(tok, text, ["ref text"], [ref], 10)
        assert n_prefix >= 40 + 2
        seq = _oracle_codes(orc, prompt, 8, **kw)
        assert np.array_equal(res[0].codes, seq[1:, prompt.shape[1]:-1]), text
        assert len(cache) == 1
    prof = ft.VoiceProfile(codes=ref, text="ref text", name="v")
    a = synth.synthesize("Yo", references=[prof], max_tokens=8, **kw)
    saved, synth._prefix_cache = synth._prefix_cache, None
    try:
        b = synth.synthesize("Yo", references=[prof], max_tokens=8, **kw)
    finally:
        synth._prefix_cache = saved
    assert a == b
    other = ft.VoiceProfile(codes=ref[:, ::-1].copy(), text="ref text", name="w")
    synth.synthesize("Yo", references=[other], max_tokens=4, **kw)
    assert len(cache) == 2
    cache.clear()
    assert len(cache) == 0


def test_synthesize_batch_equals_synthesize(tts):
    synth, tok, orc, corc, shape, cshape = tts
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    texts = ["Hi there", "Yo", "A third, longer sentence."]
    singles = [synth.synthesize(t, max_tokens=10, **kw) for t in texts]
    assert synth.synthesize_batch(texts, max_tokens=10, **kw) == singles
    assert synth.synthesize_batch([], max_tokens=10, **kw) == []


def test_synthesize_batch_on_side_by_side_streams_equals_synthesize():
    """FishTTS(max_batch=2, batch_streams=2): seven texts (more than one batch holds) go through two lock-step batches
    side by side (batch.run_batch_streams: two engines, two host threads); every WAV equals synthesize() of its text with
    the same seed - with and without a cloned voice (whose K/V prefix is built per engine and pins its utterances)."""
    import fish_tts_amd as ft
    from fish_tts_amd.tokenizer import NAMED_SPECIAL_TOKENS, ByteTokenizer
    shape = dataclasses.replace(tiny_shape(), max_seq_len=2304)
    tok = ByteTokenizer(256, NAMED_SPECIAL_TOKENS + [f"<|semantic:{i}|>" for i in range(2048)])
    synth = ft.FishTTS.synthetic(args_from_shape(shape), tok, codec_args=codec_args_from_shape(api_codec_shape()), precision="bf16",
                                 max_new_tokens=96, max_batch=2, batch_streams=2)
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)
    texts = ["Hi there", "Yo", "A third, longer sentence.", "Four", "Five is here too", "Six.", "And the seventh text"]
    singles = [synth.synthesize(t, max_tokens=10, **kw) for t in texts]          # synthesize() draws with seed 0
    assert synth.synthesize_batch(texts, max_tokens=10, seeds=[0] * len(texts), **kw) == singles
    assert len(synth._more_engines) == 1
    rng = np.random.default_rng(0)
    ref = np.concatenate([rng.integers(0, 2048, (1, 40)), rng.integers(0, 1024, (9, 40))]).astype(np.int32)
    prof = ft.VoiceProfile(codes=ref, text="the reference text", name="v")
    cloned = [synth.synthesize(t, references=[prof], max_tokens=8, **kw) for t in texts[:5]]
    assert synth.synthesize_batch(texts[:5], references=[prof], max_tokens=8, seeds=[0] * 5, **kw) == cloned
    for e in synth._more_engines:
        e.close()
    synth._engine.close()
    synth._vocoder.close()


def _write_ar_directory(tmp_path):
    """A directory laid out like the reference's checkpoint (synthesizer.py:158-197, llama.py:466-500): config.json,
    model.pth with a "state_dict" wrapper, "model." prefixes, separate wq / wk / wv, an audio_* tensor;
    tokenizer.tiktoken + special_tokens.json."""
    import base64
    import json
    from fish_tts_amd.tokenizer import NAMED_SPECIAL_TOKENS
    shape = dataclasses.replace(tiny_shape(), max_seq_len=2304)
    w = OA.random_weights(shape, seed=3)
    sd = {}
    for k, v in w.items():
        if k.endswith("attention.wqkv.weight"):
            nh, nkv, hd = (shape.fast_n_head, shape.fast_n_local_heads, shape.fast_head_dim) if k.startswith("fast_") \
                else (shape.n_head, shape.n_local_heads, shape.head_dim)
            q, kk, vv = torch.split(v, [nh * hd, nkv * hd, nkv * hd])
            pre = "model." + k[: -len("wqkv.weight")]
            sd[pre + "wq.weight"], sd[pre + "wk.weight"], sd[pre + "wv.weight"] = q.clone(), kk.clone(), vv.clone()
        else:
            sd["model." + k] = v
    sd["model.audio_projector.weight"] = torch.zeros(4, 4)
    torch.save({"state_dict": sd}, tmp_path / "model.pth")
    cfg = {k: v for k, v in args_from_shape(shape).__dict__.items()}
    cfg["model_type"] = "dual_ar"
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    ranks = {bytes([i]): i for i in range(256)}
    (tmp_path / "tokenizer.tiktoken").write_text("\n".join(f"{base64.b64encode(t).decode()} {r}" for t, r in ranks.items()))
    (tmp_path / "special_tokens.json").write_text(json.dumps(NAMED_SPECIAL_TOKENS + [f"<|semantic:{i}|>" for i in range(2048)]))
    return shape, w


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_model_directory_in_reference_layout(tmp_path, precision):
    """SURVEY §8-f F2 end to end: FishTTS(model_dir) on a directory laid out like the reference's checkpoint
    (config.json, model.pth with a "state_dict" wrapper, "model." prefixes, separate wq/wk/wv, an audio_* tensor;
    tokenizer.tiktoken + special_tokens.json; no codec.pth -> vocoder not loaded, as the reference warns)."""
    import fish_tts_amd as ft
    from fish_tts_amd.generation import generate_long
    from fish_tts_amd.prompt import build_prompt
    shape, w = _write_ar_directory(tmp_path)
    synth = ft.FishTTS(model_dir=tmp_path, precision=precision, warmup=True)     # "fp16": synthesizer.py:125-126
    try:
        assert synth._vocoder is None and synth._is_warmed_up and synth.precision == precision
        tok = synth._tokenizer
        assert tok.semantic_begin_id == shape.semantic_begin_id and tok.get_token_id("<|im_end|>") == shape.im_end_id
        kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
        res = [r for r in generate_long(engine=synth._engine, tokenizer=tok, text="Hi there", max_new_tokens=8, **kw)
               if r.action == "sample"]
        prompt = build_prompt(tok, "Hi there", None, None, 10)
        orc = OA.AROracle(shape, w, torch.float32 if precision == "fp32" else torch.float16)
        taps = []
        seq = orc.generate(torch.from_numpy(prompt), 8, frame_taps=taps, **kw).numpy()
        want = seq[1:, prompt.shape[1]:-1]
        if precision == "fp32":
            assert np.array_equal(res[0].codes, want)
        else:
            # fp16: equal up to a decision the oracle itself took by less than a few fp16 steps of its logits
            got = res[0].codes
            assert got.shape == want.shape
            diff = np.argwhere(got.T != want.T)
            if len(diff):
                f, r = int(diff[0][0]), int(diff[0][1])
                logits, _, fast = taps[f]
                l = (logits if r == 0 else fast[r - 1]).float().reshape(-1)
                top = torch.topk(l, 2).values
                assert float(top[0] - top[1]) <= 0.0075 * max(1.0, float(l.abs().max())), (f, r, got, want)
        with pytest.raises(RuntimeError, match="[Vv]ocoder"):
            synth.synthesize("Hi there", max_tokens=4)
    finally:
        synth._engine.close()


def test_model_directory_with_codec_pth(tmp_path):
    """The same directory WITH codec.pth as the reference stores it (synthesizer.py:271-286, vocoder.py:423-429, 457-463): a
    "state_dict" wrapper, every generator tensor under "generator.", the decoder's weight-normed convolutions as
    parametrizations.weight.original0 (g) / original1 (v) with g != |v|, and a discriminator tensor that must be
    dropped - at the codec widths the reference hard-codes (synthesizer.py:199-269).  FishTTS(model_dir).synthesize must
    give the PCM of the same codes decoded by an engine loaded with the plain weights, which in turn follows the f32
    oracle within the codec's bf16 tolerance."""
    import fish_tts_amd as ft
    from fish_tts_amd.codec_engine import CodecHipEngine
    from fish_tts_amd.generation import generate_long
    shape, w = _write_ar_directory(tmp_path)
    cshape = OC.CodecShape()
    cw = OC.random_weights(cshape, seed=1)
    g = torch.Generator().manual_seed(5)
    sd, n_wn = {}, 0
    for k, v in cw.items():
        if k.startswith("decoder.") and k.endswith("conv.weight") and v.dim() == 3:
            base = "generator." + k[: -len("weight")]
            stretch = 0.5 + torch.rand(v.shape[0], 1, 1, generator=g)          # v is NOT unit-norm and g is not |v|
            sd[base + "parametrizations.weight.original0"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
            sd[base + "parametrizations.weight.original1"] = v * stretch
            n_wn += 1
        else:
            sd["generator." + k] = v
    assert n_wn > 20
    sd["discriminator.convs.0.weight"] = torch.zeros(3, 3)
    torch.save({"state_dict": sd}, tmp_path / "codec.pth")
    del sd
    synth = ft.FishTTS(model_dir=tmp_path, precision="fp32", warmup=False)
    plain = None
    try:
        assert synth._vocoder is not None
        kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
        wav = synth.synthesize("Hi there", max_tokens=6, **kw)
        codes = [r.codes for r in generate_long(engine=synth._engine, tokenizer=synth._tokenizer, text="Hi there", max_new_tokens=6, **kw)
                 if r.action == "sample"][0]
        with wave.open(io.BytesIO(wav), "rb") as wf:
            pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
        assert pcm.shape == (codes.shape[1] * cshape.frame_len,)
        plain = CodecHipEngine(device=0, max_frames=64)
        plain.load_state_dict(cw)
        audio = plain.decode(codes[None].astype(np.int64))[0]
        # the fold g * v / |v| is f32 arithmetic on the host: the folded weights equal the plain ones to f32 rounding, the
        # bf16 operand copies then agree except where that rounding crosses a bf16 tie
        ref_pcm = (np.clip(audio, -1, 1) * 32767).astype(np.int16).astype(np.float32)
        fold_err = np.sqrt(np.mean((pcm.astype(np.float32) - ref_pcm) ** 2)) / np.sqrt(np.mean(ref_pcm ** 2))
        assert fold_err <= 1e-2, fold_err
        want, _ = OC.CodecOracle(cshape, cw).decode(torch.from_numpy(codes.astype(np.int64))[None], torch.tensor([codes.shape[1]]))
        want = want[0, 0].numpy()
        err = np.sqrt(np.mean((audio - want) ** 2)) / np.sqrt(np.mean(want ** 2))
        assert err <= 3e-2, err
    finally:
        synth._engine.close()
        if synth._vocoder is not None:
            synth._vocoder.close()
        if plain is not None:
            plain.close()


def test_seamless_stream_equals_the_full_decode(tts):
    """synthesize_stream(seamless=True): the chunks, concatenated, are exactly the PCM of decoding the whole code
    sequence at once (strict causality of the codec), unlike the reference's independent chunk decodes."""
    synth, tok, orc, corc, shape, cshape = tts
    from fish_tts_amd.generation import generate_long
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    chunks = list(synth.synthesize_stream("Hi there", chunk_tokens=6, min_first_chunk=3, max_tokens=20, seamless=True, **kw))
    codes = np.concatenate([r.codes for r in generate_long(engine=synth._engine, tokenizer=tok, text="Hi there",
                                                           max_new_tokens=20, streaming=True, **kw) if r.action == "sample"], axis=1)
    whole = synth._decode_to_pcm(codes)
    assert b"".join(chunks) == whole
    plain = list(synth.synthesize_stream("Hi there", chunk_tokens=6, min_first_chunk=3, max_tokens=20, **kw))
    assert [len(c) for c in plain] == [len(c) for c in chunks] and b"".join(plain) != whole
