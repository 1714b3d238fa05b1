"""Teacher-forcing fixture at the real openaudio-s1-mini shapes (28+4 layers, V = 155 776): the REFERENCE (imported as in
make_golden.py) generates N_NEW greedy frames with repetition_penalty = 1.0 in fp32 AND in bf16 on seeded synthetic
weights (regenerated from the seed by the tests); for EVERY decision of every frame the oracle's top-1/top-2 margin is
recorded, and the slow logits' top-8 of every frame.  tests/test_ar_gpu.py prefills prompt + golden[:k] for every k and
judges frame k's eleven decisions each against its own margin (fp32: index-exact).

With penalty 1.0 the frame a prefill call yields (inference.py:353-362, no penalty) and the frame the decode loop
yields at the same position are the same function of the same tokens, so every golden frame can be checked alone.
Run here only: `python tests/golden/make_golden_s1mini_tf.py` (~6 min, ~12 GB)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import build_reference_model, import_reference  # noqa: E402
from tests.shapes import make_prompt, s1mini_shape  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
STD, SEED_W, LP, N_NEW = 0.02, 0, 24, 17
LOUD = (16, 4.0)     # oracle.ar.random_weights: a few loud head rows give the decisions a trained model's margins
KW = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.0)


def main():
    torch.set_num_threads(8)
    from oracle import ar as O
    llama, inference, _ = import_reference()
    shape = s1mini_shape()
    w = O.random_weights(shape, seed=SEED_W, std=STD, loud=LOUD)
    prompt = make_prompt(shape, LP, seed=1, n_vq=3)
    out = {"prompt": prompt.numpy(), "std": np.float32(STD), "seed_w": np.int64(SEED_W), "n_new": np.int64(N_NEW),
           "loud_n": np.int64(LOUD[0]), "loud_factor": np.float32(LOUD[1])}
    for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        model, tok = build_reference_model(llama, shape, w, dtype)
        seq = inference.generate(model=model, prompt=prompt.clone(), max_new_tokens=N_NEW, audio_masks=None,
                                 audio_parts=None, **KW)
        out[f"{tag}.seq"] = seq.numpy().copy()
        del model
        orc = O.AROracle(shape, w, dtype)
        taps = []
        want = orc.generate(prompt.clone(), N_NEW, frame_taps=taps, **KW).numpy()
        assert np.array_equal(want, out[f"{tag}.seq"]), f"oracle != reference at s1-mini shapes ({tag})"
        margins = np.zeros((len(taps), shape.num_codebooks), dtype=np.float32)
        scale = np.zeros((len(taps), shape.num_codebooks), dtype=np.float32)      # largest |logit| of each decision's own vector
        top = np.zeros((len(taps), 8), dtype=np.int64)
        topv = np.zeros((len(taps), 8), dtype=np.float32)
        for f, (logits, _, fast) in enumerate(taps):
            l = logits.float().reshape(-1)
            tk = torch.topk(l, 8)
            top[f], topv[f] = tk.indices.numpy(), tk.values.numpy()
            margins[f, 0] = float(tk.values[0] - tk.values[1])
            scale[f, 0] = float(l.abs().max())
            for c in range(1, shape.num_codebooks):
                fl = fast[c - 1].float().reshape(-1)
                t2 = torch.topk(fl, 2).values
                margins[f, c] = float(t2[0] - t2[1])
                scale[f, c] = float(fl.abs().max())
        out[f"{tag}.margins"] = margins
        out[f"{tag}.scale"] = scale
        out[f"{tag}.slow_top8"] = top
        out[f"{tag}.slow_top8_logits"] = topv
        out[f"{tag}.logit_absmax"] = np.float32(max(float(t[0].float().abs().max()) for t in taps))
        del orc
        print(tag, out[f"{tag}.seq"][:, LP:LP + 4], "min margin", margins.min(), flush=True)
    np.savez_compressed(os.path.join(OUT, "ar_s1mini_tf.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
