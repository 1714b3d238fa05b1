"""The persistent frame engine under PRODUCT conditions (it is the default batch-1 path of synthesize / synthesize_stream):

* behind the public API at the s1-mini widths with the real codec widths, where the stream's decoder thread runs
  ft_codec_decode on its own stream WHILE the AR loop replays engine frames (fish_tts/synthesizer.py:483-584: two host
  threads drive the GPU);
* the same collision without the API: a thread looping 215-frame codec decodes beside 200 engine frames;
* a hand-off time-out (injected: one workgroup publishes nothing) is recovered inside the call - the frames come out as
  the launch path's, the next call runs on the engine again, the second time-out turns the engine off for the context;
* a second AR context on the same device does not get the engine (two sets of one-workgroup-per-CU launches would
  starve each other).

Every comparison is bit for bit against the launch path (FT_NO_ENGINE), which tests/test_ar_gpu.py pins to the oracle
and the reference's golden vectors.  reference: fish_tts/models/inference.py:83-276, fish_tts/synthesizer.py:431-584."""
import dataclasses
import threading

import numpy as np
import pytest

from tests.hip_util import args_from_shape, make_pair
from tests.shapes import make_prompt
from tests.test_ar_gpu import medium_shape

pytestmark = pytest.mark.gpu


def _tts(monkeypatch, engine_on: bool, max_new_tokens: int = 160):
    import fish_tts_amd as ft
    from fish_tts_amd.tokenizer import NAMED_SPECIAL_TOKENS, ByteTokenizer
    monkeypatch.delenv("FT_NO_ENGINE", raising=False)
    if not engine_on:
        monkeypatch.setenv("FT_NO_ENGINE", "1")
    # generate_long refuses prompts longer than max_seq_len - 2048 (inference.py:794)
    shape = dataclasses.replace(medium_shape(), max_seq_len=2048 + 256)
    tok = ByteTokenizer(1000, NAMED_SPECIAL_TOKENS + [f"<|semantic:{i}|>" for i in range(4096)])
    assert tok.semantic_begin_id == shape.semantic_begin_id and tok.get_token_id("<|im_end|>") == shape.im_end_id
    # the codec at its REAL widths (CodecArgs defaults = synthesizer.py:199-269), random weights
    synth = ft.FishTTS.synthetic(args_from_shape(shape), tok, precision="bf16", seed=0, warmup=False,
                                 max_new_tokens=max_new_tokens, std=0.05)
    return synth


def _close(synth):
    synth._engine.close()
    synth._vocoder.close()


def test_stream_api_engine_beside_codec_thread_equals_launch_path(monkeypatch):
    """FishTTS.synthesize_stream in bf16 at the s1-mini widths: the engine is on (flags == 3), the decoder worker decodes
    10- and 20-frame chunks at the real codec widths while the AR loop keeps replaying engine frames.  Every PCM chunk must
    equal the FT_NO_ENGINE run's, no hand-off may time out."""
    kw = dict(chunk_tokens=20, min_first_chunk=10, max_tokens=120, temperature=0.7, top_p=0.8, repetition_penalty=1.1)
    runs = []
    for engine_on in (True, False):
        synth = _tts(monkeypatch, engine_on)
        flags = synth._engine.engine_state()[0]
        assert flags == (3 if engine_on else 0), synth._engine.frame_path()
        chunks = list(synth.synthesize_stream("The quick brown fox jumps over the lazy dog.", **kw))
        wav = synth.synthesize("Streaming and batch share one engine.", max_tokens=40)
        flags2, aborted, where = synth._engine.engine_state()
        assert aborted == 0 and flags2 == flags, (aborted, where, synth._engine.frame_path())
        runs.append((chunks, wav))
        _close(synth)
    (a, wa), (b, wb) = runs
    assert len(a) == len(b) and len(a) >= 2, (len(a), len(b))
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, f"PCM chunk {i} differs between the engine and the launch path"
    assert wa == wb


def test_engine_frames_beside_a_looping_codec_decode(monkeypatch):
    """The collision without the API: one thread decodes (10, 215) codes at the real codec widths in a loop on the codec
    context's stream while this thread runs 200 frames on the engine.  Frames == an idle-chip engine run == the launch path;
    no time-out."""
    from fish_tts_amd.codec_engine import CodecHipEngine
    shape = dataclasses.replace(medium_shape(), max_seq_len=1024)
    prompt = make_prompt(shape, 40, seed=12, n_vq=3).numpy()
    n = 200

    def run(engine_on, with_codec):
        monkeypatch.delenv("FT_NO_ENGINE", raising=False)
        if not engine_on:
            monkeypatch.setenv("FT_NO_ENGINE", "1")
        eng, _ = make_pair(shape, "bf16", max_new_tokens=n + 8, std=0.05)
        assert eng.engine_state()[0] == (3 if engine_on else 0)
        stop, err, count = threading.Event(), [], [0]
        th, codec = None, None
        if with_codec:
            codec = CodecHipEngine.synthetic(device=0, max_frames=215, seed=1)
            g = np.random.default_rng(0)
            codes = np.concatenate([g.integers(0, 4096, (1, 215)), g.integers(0, 1024, (9, 215))]).astype(np.int32)

            def loop():
                try:
                    ref = codec.decode(codes)
                    while not stop.is_set():
                        out = codec.decode(codes)
                        assert np.array_equal(out, ref)          # the codec's own result is unaffected, too
                        count[0] += 1
                except Exception as e:  # noqa: BLE001
                    err.append(e)
            th = threading.Thread(target=loop, daemon=True)
            th.start()
            while count[0] < 1 and not err:          # the loop is really running before the frames start
                pass
        sp = eng._sampling(0.7, 0.8, 1.1, seed=5, ban_eos=True)
        first = eng.prefill(prompt, sp, slot=0)
        frames, cnt = eng.decode(n, [sp], poll=50)
        if th is not None:
            stop.set()
            th.join()
            codec.close()
        _, aborted, where = eng.engine_state()
        eng.close()
        assert not err, err
        assert aborted == 0, where
        return first, frames[0, : cnt[0]].copy(), count[0]
    f0, a, _ = run(False, False)
    f1, b, _ = run(True, False)
    f2, c, decodes = run(True, True)
    assert decodes >= 2, decodes            # codec decodes really overlapped the 200 frames
    assert len(a) == n
    assert np.array_equal(f0, f1) and np.array_equal(f0, f2)
    assert np.array_equal(a, b)
    assert np.array_equal(a, c), int(np.argmax((a != c).any(axis=1)))


def test_hand_off_time_out_is_recovered_in_process(monkeypatch):
    """ft_test_engine_fault makes one workgroup of the next engine launch publish nothing: the launch times out (200 ms),
    ft_ar_decode clears the engine's control words and buffers, redoes the burst on the launch path and returns the launch
    path's frames; the NEXT call runs on the engine again.  A second time-out (in the codebook loop of a prefill's first
    frame this time) turns the engine off for the context; results stay those of the launch path throughout."""
    shape = dataclasses.replace(medium_shape(), max_seq_len=1024)
    prompt = make_prompt(shape, 30, seed=3, n_vq=2).numpy()
    prompt2 = make_prompt(shape, 21, seed=8, n_vq=1).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)

    monkeypatch.setenv("FT_NO_ENGINE", "1")
    ref, _ = make_pair(shape, "bf16", max_new_tokens=80, std=0.05)
    sp = ref._sampling(seed=9, ban_eos=True, **kw)
    want_first = ref.prefill(prompt, sp, slot=0)
    want, wn = ref.decode(48, [sp], poll=8)
    want2_first = ref.prefill(prompt2, sp, slot=0)
    want2, wn2 = ref.decode(16, [sp], poll=8)
    ref.close()

    monkeypatch.delenv("FT_NO_ENGINE", raising=False)
    eng, _ = make_pair(shape, "bf16", max_new_tokens=80, std=0.05)
    assert eng.engine_state()[0] == 3
    first = eng.prefill(prompt, sp, slot=0)
    assert np.array_equal(first, want_first)
    a, na = eng.decode(16, [sp], poll=8)                      # clean engine frames
    eng.inject_engine_fault(which=0, workgroup=5)
    b, nb_ = eng.decode(16, [sp], poll=8)                     # first burst of this call times out and is redone
    flags, strikes, _ = eng.engine_state()
    assert (flags, strikes) == (3, 1), (flags, strikes, eng.frame_path())
    c, nc = eng.decode(16, [sp], poll=8)                      # on the engine again
    assert eng.engine_state()[:2] == (3, 1)
    got = np.concatenate([a[0, : na[0]], b[0, : nb_[0]], c[0, : nc[0]]])
    assert np.array_equal(got, want[0, : wn[0]]), int(np.argmax((got != want[0, : wn[0]]).any(axis=1)))
    # second strike: the codebook loop of the next prefill's first frame
    eng.inject_engine_fault(which=1, workgroup=200)
    first2 = eng.prefill(prompt2, sp, slot=0)
    assert np.array_equal(first2, want2_first)
    flags, strikes, _ = eng.engine_state()
    assert (flags, strikes) == (0, 2), (flags, strikes)
    assert "turned off" in eng.frame_path()
    d, nd = eng.decode(16, [sp], poll=8)
    assert np.array_equal(d[0, : nd[0]], want2[0, : wn2[0]])
    eng.close()


def test_time_out_in_a_later_burst_and_in_first_frames(monkeypatch):
    """The recovery's other entrances: (1) a slow-stack time-out placed after ten passing launches - frame 11 of
    decode(24, poll=8), i.e. the SECOND burst, inside a four-frame graph: the slot is rewound to the burst's first frame
    (position, frame count, input column from the frame store) and the burst redone on the launch path; (2) a codebook-loop
    time-out in ft_ar_first_frames(n = 1) after ft_ar_prefill_slow (prefill_many).  Frames equal the launch path's bit for
    bit throughout; the second strike releases the engine, and another context of the process may then take it."""
    shape = dataclasses.replace(medium_shape(), max_seq_len=1024)
    prompt = make_prompt(shape, 26, seed=13, n_vq=2).numpy()
    prompt2 = make_prompt(shape, 17, seed=18, n_vq=1).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)

    monkeypatch.setenv("FT_NO_ENGINE", "1")
    ref, _ = make_pair(shape, "bf16", max_new_tokens=64, std=0.05)
    sp = ref._sampling(seed=21, ban_eos=True, **kw)
    want_first = ref.prefill(prompt, sp, slot=0)
    want, wn = ref.decode(24, [sp], poll=8)
    want2_first = ref.prefill_many([prompt2], [sp])[0]
    want2, wn2 = ref.decode(8, [sp], poll=8)
    ref.close()

    monkeypatch.delenv("FT_NO_ENGINE", raising=False)
    eng, _ = make_pair(shape, "bf16", max_new_tokens=64, std=0.05)
    assert eng.engine_state()[0] == 3
    assert np.array_equal(eng.prefill(prompt, sp, slot=0), want_first)
    eng.inject_engine_fault(which=0, workgroup=77, skip=10)
    got, gn = eng.decode(24, [sp], poll=8)
    assert eng.engine_state()[:2] == (3, 1), (eng.engine_state(), eng.frame_path())
    assert gn[0] == wn[0] and np.array_equal(got[0, : gn[0]], want[0, : wn[0]]), int(np.argmax((got[0] != want[0]).any(axis=1)))
    eng.inject_engine_fault(which=1, workgroup=3)
    first2 = eng.prefill_many([prompt2], [sp])[0]
    assert np.array_equal(first2, want2_first)
    flags, strikes, _ = eng.engine_state()
    assert (flags, strikes) == (0, 2) and "turned off" in eng.frame_path()
    d, nd = eng.decode(8, [sp], poll=8)
    assert np.array_equal(d[0, : nd[0]], want2[0, : wn2[0]])
    other, _ = make_pair(shape, "bf16", max_new_tokens=16, std=0.05)      # the seat was released at the second strike
    assert other.engine_state()[0] == 3, other.frame_path()
    other.close()
    eng.close()


def test_second_context_on_the_device_keeps_the_launch_path(monkeypatch):
    """One engine context per device and process: a second AR context takes the launch path (and says why); when the first
    is closed, a new one gets the engine."""
    monkeypatch.delenv("FT_NO_ENGINE", raising=False)
    shape = dataclasses.replace(medium_shape(), max_seq_len=512)
    a, _ = make_pair(shape, "bf16", max_new_tokens=16)
    b, _ = make_pair(shape, "bf16", max_new_tokens=16)
    assert a.engine_state()[0] == 3 and b.engine_state()[0] == 0
    assert "another context" in b.frame_path(), b.frame_path()
    prompt = make_prompt(shape, 12, seed=2, n_vq=1).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1, seed=4)
    assert np.array_equal(a.generate(prompt, 8, **kw), b.generate(prompt, 8, **kw))
    a.close()
    c, _ = make_pair(shape, "bf16", max_new_tokens=16)
    assert c.engine_state()[0] == 3
    b.close()
    c.close()
