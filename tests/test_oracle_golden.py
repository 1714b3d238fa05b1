"""Pins the oracle (oracle/ar.py) against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ar as O
from tests.shapes import tiny_shape, tiny_shape_b

G = os.path.join(os.path.dirname(__file__), "golden")

CASES = [("greedy_rep1.0", dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.0), None),
         ("greedy_rep1.1", dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1), None),
         ("sampled_seed7", dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1), 7),
         ("sampled_seed11", dict(temperature=1.0, top_p=0.95, repetition_penalty=1.5), 11)]


@pytest.mark.parametrize("name,shape_fn,dtype,n_new", [
    ("ar_tiny_f32", tiny_shape, torch.float32, 16),
    ("ar_tiny_bf16", tiny_shape, torch.bfloat16, 16),
    ("ar_tinyb_f32", tiny_shape_b, torch.float32, 12),
    ("ar_tinyb_bf16", tiny_shape_b, torch.bfloat16, 12),
    ("ar_tiny_f16", tiny_shape, torch.float16, 16),          # precision="fp16" (synthesizer.py:125-126)
    ("ar_tinyb_f16", tiny_shape_b, torch.float16, 12),
])
def test_ar_oracle_matches_reference(name, shape_fn, dtype, n_new):
    torch.set_num_threads(4)
    gold = np.load(os.path.join(G, name + ".npz"))
    shape = shape_fn()
    orc = O.AROracle(shape, O.random_weights(shape, seed=0), dtype)
    prompt = torch.from_numpy(gold["prompt"])
    for cname, kw, seed in CASES:
        orc.reset()
        if seed is not None:
            torch.manual_seed(seed)
        seq = orc.generate(prompt.clone(), n_new, **kw)
        assert np.array_equal(seq.numpy(), gold[f"{cname}.seq"]), cname
        orc.reset()
        if seed is not None:
            torch.manual_seed(seed)
        cols = orc.generate_stream(prompt.clone(), n_new, **kw)
        assert np.array_equal(torch.cat(cols, dim=1).numpy(), gold[f"{cname}.stream"]), cname
    # frame-0 logits/hidden: bit-equal (same ops, same order)
    orc.reset()
    with torch.inference_mode():
        logits, hidden = orc.slow_forward(prompt.view(1, shape.num_codebooks + 1, -1), torch.arange(prompt.size(1)))
    assert np.array_equal(logits.float().numpy().reshape(-1), gold["frame0.logits"])
    assert np.array_equal(hidden.float().numpy().reshape(-1), gold["frame0.hidden"])


def test_sampling_oracle_matches_reference():
    gold = np.load(os.path.join(G, "sampling.npz"))
    for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        for V in (1024, 2320):
            logits = torch.from_numpy(gold[f"{tag}.V{V}.logits"]).to(dtype)
            prev = torch.from_numpy(gold[f"{tag}.V{V}.prev"])
            for tp in (0.8, 0.2, 1.0, 1e-6):
                for rep in (1.0, 1.1, 1.5):
                    l = logits.clone()
                    probs = O.logits_to_probs(l, torch.tensor(0.7), torch.tensor(tp), torch.tensor(rep), prev)
                    key = f"{tag}.V{V}.tp{tp}.rep{rep}"
                    assert np.array_equal(probs.float().numpy(), gold[key + ".probs"]), key
                    assert np.array_equal(l.float().numpy(), gold[key + ".penalised"]), key


def test_batch_drops_last_column_stream_keeps_it():
    """inference.py:839 vs 721/271: batch mode returns n-1 usable frames, streaming n."""
    gold = np.load(os.path.join(G, "ar_tiny_f32.npz"))
    T = gold["prompt"].shape[1]
    seq, stream = gold["greedy_rep1.1.seq"], gold["greedy_rep1.1.stream"]
    assert seq.shape[1] - T == stream.shape[1]
    assert np.array_equal(seq[1:, T:], stream)


def test_ar_oracle_matches_reference_at_s1mini_shapes():
    """G6: the real model shapes (28+4 layers, V = 155 776), bf16, seeded weights regenerated here (~40 s, 6 GB)."""
    from tests.shapes import s1mini_shape
    g = np.load(os.path.join(G, "ar_s1mini.npz"))
    shape = s1mini_shape()
    w = O.random_weights(shape, seed=int(g["seed_w"]), std=float(g["std"]))
    orc = O.AROracle(shape, w, torch.bfloat16)
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    prompt = torch.from_numpy(g["prompt"])
    got = orc.generate(prompt.clone(), int(g["n_new"]), **kw).numpy()
    assert np.array_equal(got, g["bf16.seq"])
    cols = list(orc.generate_stream(prompt.clone(), int(g["n_new"]), **kw))
    assert np.array_equal(torch.cat(cols, dim=1).numpy(), g["bf16.stream"])


def test_oracle_pinned_at_s1mini_shapes_by_reference_generated_frames():
    """The oracle at the REAL shapes (28 + 4 layers, vocabulary 155 776) against tests/golden/ar_s1mini_tf.npz, 17 greedy
    frames the imported reference generated (make_golden_s1mini_tf.py): fp32 index-exact for every frame; bf16 equal up to
    the first decision whose reference margin is below the GPU tests' tolerance (0.03 x logit range: CPU bf16 kernels may
    re-associate sums with the thread count, and a later frame depends on every earlier decision)."""
    import os
    import numpy as np
    import torch
    from oracle import ar as O
    from tests.shapes import s1mini_shape
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ar_s1mini_tf.npz"))
    shape = s1mini_shape()
    w = O.random_weights(shape, seed=int(g["seed_w"]), std=float(g["std"]), loud=(int(g["loud_n"]), float(g["loud_factor"])))
    prompt = torch.from_numpy(g["prompt"])
    Lp, n = prompt.shape[1], int(g["n_new"])
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.0)
    seq = O.AROracle(shape, w, torch.float32).generate(prompt.clone(), n, **kw).numpy()
    assert np.array_equal(seq, g["f32.seq"])
    seq = O.AROracle(shape, w, torch.bfloat16).generate(prompt.clone(), n, **kw).numpy()
    want, margins = g["bf16.seq"], g["bf16.margins"]
    tol = 0.03 * 2 * float(g["bf16.logit_absmax"])
    for f in range(min(seq.shape[1], want.shape[1]) - Lp):
        col_g, col_w = seq[:, Lp + f], want[:, Lp + f]
        if np.array_equal(col_g, col_w):
            continue
        first = int(np.argmax(col_g != col_w))                  # row 0 = semantic token, row c = codebook c - 1
        dec = 0 if first == 0 else first - 1                    # decision index inside the frame (semantic draw = 0)
        assert margins[f, min(dec, margins.shape[1] - 1)] < tol, (f, first, margins[f])
        break
