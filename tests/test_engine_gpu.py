"""The persistent frame engine (csrc/frame_engine.h) against the launch path it replaces: same weights, same prompt,
same sampling -> the token matrices, the vocabulary logits and the hidden state must be equal BIT FOR BIT (every phase
of the engine reproduces the per-row arithmetic of the launch-path kernels), and both are pinned to the oracle by
tests/test_ar_gpu.py.  A stale hand-off inside the engine shows up here as a difference.

reference path: fish_tts/models/inference.py:83-155 (one frame), llama.py:400-453, 561-580."""
import dataclasses

import numpy as np
import pytest

from tests.hip_util import NoiseTape, make_pair
from tests.shapes import make_prompt, s1mini_shape
from tests.test_ar_gpu import medium_shape

pytestmark = pytest.mark.gpu


def _run(monkeypatch, shape, engine_on, prompt, n_new, kw, tape=None, precision="bf16", env=()):
    """engine_on: False = launch path, True = both engines, "slow" = the slow-stack engine only; env: switches that turn
    single engine features off (FT_NO_PAIR, FT_NO_QKV0, FT_NO_RELAY)."""
    monkeypatch.delenv("FT_NO_ENGINE", raising=False)
    monkeypatch.delenv("FT_NO_FAST_ENGINE", raising=False)
    for k in ("FT_NO_PAIR", "FT_NO_QKV0", "FT_NO_RELAY"):
        monkeypatch.delenv(k, raising=False)
    for k in env:
        monkeypatch.setenv(k, "1")
    if not engine_on:
        monkeypatch.setenv("FT_NO_ENGINE", "1")
    elif engine_on == "slow":
        monkeypatch.setenv("FT_NO_FAST_ENGINE", "1")
    eng, _ = make_pair(shape, precision, max_new_tokens=n_new + 8)
    flags, _, _ = eng.engine_state()
    if tape is not None:
        eng.set_noise(tape.table())
    seq = eng.generate(prompt, n_new, **kw)
    logits, hidden = eng.debug_state()
    _, aborted, where = eng.engine_state()
    eng.close()
    assert aborted == 0, where
    return flags, seq, logits, hidden


@pytest.mark.parametrize("no_xl", [False, True])
@pytest.mark.parametrize("max_seq_len,Lp", [(512, 40), (1024, 40), (1024, 300), (4096, 900)])
def test_engine_frames_equal_launch_frames_greedy(monkeypatch, max_seq_len, Lp, no_xl):
    """2+2 layers at the s1-mini widths.  On a chip of 8 XCDs x 32 CUs the slow stack runs its XCD-local form (one kv head
    per XCD, 32 KV splits at every context length, launches and engine alike); with FT_NO_XL (and on other chips) the split
    count follows the cache size: max_seq_len 512 -> one KV split per head, 1024 -> 8 splits merged by the attention
    workgroups, 4096 with a 900-token prompt -> 16 splits (two merge chunks)."""
    if no_xl:
        monkeypatch.setenv("FT_NO_XL", "1")
    else:
        monkeypatch.delenv("FT_NO_XL", raising=False)
    shape = dataclasses.replace(medium_shape(), max_seq_len=max_seq_len)
    prompt = make_prompt(shape, Lp, seed=4, n_vq=4).numpy()
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    fa, a, la, ha = _run(monkeypatch, shape, False, prompt, 24, kw)
    for mode, want in (("slow", 1), (True, 3)):
        fb, b, lb, hb = _run(monkeypatch, shape, mode, prompt, 24, kw)
        assert fa == 0 and fb == want, (fa, fb)          # the run really took the engine(s)
        assert np.array_equal(a, b), mode
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), mode
        assert np.array_equal(ha.view(np.uint32), hb.view(np.uint32)), mode


@pytest.mark.parametrize("over,want_flags", [
    (dict(intermediate_size=4096, fast_intermediate_size=4096), 3),     # the feed-forward width SURVEY could not verify: both engines
    (dict(intermediate_size=4096), 3),                                   # slow stack 4096, fast stack 3072
    (dict(n_local_heads=4), 3),                                          # four kv heads: the general split form of the attention
    (dict(n_local_heads=4, intermediate_size=4096), 0),                  # no instantiation: launch path, and it says why
])
@pytest.mark.parametrize("max_seq_len,Lp", [(1024, 40), (1024, 300)])
def test_engine_shape_classes_equal_launch_frames(monkeypatch, over, want_flags, max_seq_len, Lp):
    """The engine kernels are instantiated for a short list of shape classes (csrc/engine.hip: eng_slow_shapes /
    eng_fast_shapes; llama.py:74-86 reads the widths from config.json): every class must give the launch path's frames,
    vocabulary logits and hidden state bit for bit, greedy and sampled; widths outside the list keep the launch path."""
    monkeypatch.delenv("FT_NO_XL", raising=False)
    shape = dataclasses.replace(medium_shape(**over), max_seq_len=max_seq_len)
    prompt = make_prompt(shape, Lp, seed=14, n_vq=4).numpy()
    for kw, tape in ((dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1), None),
                     (dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1), NoiseTape(shape, 24, seed=5))):
        fa, a, la, ha = _run(monkeypatch, shape, False, prompt, 16, kw, tape)
        fb, b, lb, hb = _run(monkeypatch, shape, True, prompt, 16, kw, tape)
        assert fa == 0 and fb == want_flags, (fa, fb)
        assert np.array_equal(a, b)
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
        assert np.array_equal(ha.view(np.uint32), hb.view(np.uint32))


def test_engine_frames_equal_launch_frames_sampled(monkeypatch):
    shape = dataclasses.replace(medium_shape(), max_seq_len=1024)
    prompt = make_prompt(shape, 24, seed=6, n_vq=3).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)
    tape = NoiseTape(shape, 40, seed=3)
    _, a, la, _ = _run(monkeypatch, shape, False, prompt, 32, kw, tape)
    for mode, want in (("slow", 1), (True, 3)):
        fb, b, lb, _ = _run(monkeypatch, shape, mode, prompt, 32, kw, tape)
        assert fb == want
        assert np.array_equal(a, b), mode
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), mode


@pytest.mark.parametrize("env", [("FT_NO_PAIR",), ("FT_NO_QKV0",), ("FT_NO_PAIR", "FT_NO_QKV0"), ("FT_NO_RELAY",)])
def test_engine_feature_switches_keep_the_bits(monkeypatch, env):
    """The fast loop's paired first pass (positions 0 and 1 as two rows), its layer-0 q k v table and the per-XCD relay are
    each optional (LDS budget, two-codebook models, FT_NO_*): with any of them off the frames are still the launch path's."""
    shape = dataclasses.replace(medium_shape(), max_seq_len=1024)
    prompt = make_prompt(shape, 33, seed=9, n_vq=2).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)
    tape = NoiseTape(shape, 32, seed=5)
    _, a, la, ha = _run(monkeypatch, shape, False, prompt, 20, kw, tape)
    fb, b, lb, hb = _run(monkeypatch, shape, True, prompt, 20, kw, tape, env=env)
    assert fb == 3
    assert np.array_equal(a, b), env
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), env
    assert np.array_equal(ha.view(np.uint32), hb.view(np.uint32)), env


@pytest.mark.parametrize("over", [dict(num_codebooks=2), dict(num_codebooks=3), dict(num_codebooks=7, n_fast_layer=1),
                                  dict(n_fast_layer=3, fast_attention_qk_norm=True), dict(n_layer=1, num_codebooks=4)])
def test_engine_other_depths_and_codebook_counts(monkeypatch, over):
    """Edges of the fast loop's structure: two codebooks (the paired first pass is the whole loop, no table), three (one
    table step), one / three fast layers (the row-0 tail skip sits on the last layer), q/k norms inside the fast attention,
    a one-layer slow stack.  Each must still equal the launch path bit for bit."""
    shape = dataclasses.replace(medium_shape(**over), max_seq_len=1024)
    prompt = make_prompt(shape, 17, seed=21, n_vq=min(2, shape.num_codebooks)).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)
    tape = NoiseTape(shape, 24, seed=8)
    _, a, la, ha = _run(monkeypatch, shape, False, prompt, 14, kw, tape)
    fb, b, lb, hb = _run(monkeypatch, shape, True, prompt, 14, kw, tape)
    assert fb == 3, fb
    assert np.array_equal(a, b), over
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), over
    assert np.array_equal(ha.view(np.uint32), hb.view(np.uint32)), over


def test_engine_long_run_crosses_the_tag_wrap(monkeypatch):
    """Hand-off tags are 15 bits of an epoch that advances by 64 per launch (two launches per frame): they repeat every
    256 frames.  700 sampled frames (three wraps) on the engine must still equal the launch path's, frame for frame - a
    granule left over from 256 frames earlier would carry a valid-looking tag."""
    shape = dataclasses.replace(medium_shape(), max_seq_len=1024)
    prompt = make_prompt(shape, 20, seed=11, n_vq=2).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)
    n = 700

    def run(engine_on):
        monkeypatch.delenv("FT_NO_ENGINE", raising=False)
        if not engine_on:
            monkeypatch.setenv("FT_NO_ENGINE", "1")
        eng, _ = make_pair(shape, "bf16", max_new_tokens=n + 8)
        flags = eng.engine_state()[0]
        sp = eng._sampling(0.7, 0.8, 1.1, seed=7, ban_eos=True)       # <|im_end|> masked: the run has its full length
        first = eng.prefill(prompt, sp, slot=0)
        frames, cnt = eng.decode(n, [sp], poll=64)
        _, aborted, where = eng.engine_state()
        eng.close()
        assert aborted == 0, where
        return flags, first, frames[0, : cnt[0]]
    fa, a0, a = run(False)
    fb, b0, b = run(True)
    assert fa == 0 and fb == 3
    assert len(a) == n and len(b) == n
    assert np.array_equal(a0, b0)
    assert np.array_equal(a, b), int(np.argmax((a != b).any(axis=1)))


def test_engine_full_depth_equals_launch_path(monkeypatch):
    """28 + 4 layers, vocabulary 155 776 (BASELINE configs[1] shapes), sampled: 16 frames."""
    shape = s1mini_shape(max_seq_len=1024)
    prompt = make_prompt(shape, 48, seed=1, n_vq=0).numpy()
    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1)
    _, a, la, ha = _run(monkeypatch, shape, False, prompt, 16, kw)
    fb, b, lb, hb = _run(monkeypatch, shape, True, prompt, 16, kw)
    assert fb == 3
    assert np.array_equal(a, b)
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
    assert np.array_equal(ha.view(np.uint32), hb.view(np.uint32))


def test_other_configurations_keep_the_launch_path(monkeypatch):
    """f32 precision and the tiny widths are outside the engine's shape class: flags stay 0 and nothing changes."""
    from tests.shapes import tiny_shape
    monkeypatch.delenv("FT_NO_ENGINE", raising=False)
    for shape, precision in ((tiny_shape(), "bf16"), (medium_shape(), "fp32")):
        eng, _ = make_pair(shape, precision)
        assert eng.engine_state()[0] == 0
        eng.close()
