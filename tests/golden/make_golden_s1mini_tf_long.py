"""Teacher-forcing fixtures at the real openaudio-s1-mini shapes at the cached positions the bench and configs[4]
actually decode at: the REFERENCE (imported as in make_golden.py) generates N_NEW greedy frames (repetition penalty 1.0)
in fp32 AND bf16 after a prompt of ~250 positions (the end of a 10 s utterance: bench positions 48..263) and of ~780
positions (a 30 s voice-cloning reference + text: configs[4], positions 777..992), mostly VQ columns as such prompts
are.  Same record per decision as make_golden_s1mini_tf.py (top-1/top-2 margin, the decision's own logit range, the slow
logits' top-8); tests/test_ar_gpu.py prefills prompt + golden[:k] and judges frame k from the prompt pass and frame k+1
from one decode-loop step (bf16: the persistent frame engine; launch path: split-KV decode attention at long context).

Run here only: `python tests/golden/make_golden_s1mini_tf_long.py` (~10 min, ~14 GB)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import build_reference_model, import_reference  # noqa: E402
from tests.shapes import make_prompt, s1mini_shape  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
STD, SEED_W, N_NEW = 0.02, 0, 8
LOUD = (16, 4.0)
BLOCKS = (("p250", 250, 200, 11), ("p780", 780, 700, 12))     # tag, prompt positions, VQ columns among them, prompt seed
KW = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.0)
MAX_SEQ = 1024


def record(shape, taps, out, key):
    margins = np.zeros((len(taps), shape.num_codebooks), dtype=np.float32)
    scale = np.zeros((len(taps), shape.num_codebooks), dtype=np.float32)
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
    out[f"{key}.margins"] = margins
    out[f"{key}.scale"] = scale
    out[f"{key}.slow_top8"] = top
    out[f"{key}.slow_top8_logits"] = topv
    out[f"{key}.logit_absmax"] = np.float32(max(float(t[0].float().abs().max()) for t in taps))
    return margins


def main():
    torch.set_num_threads(8)
    from oracle import ar as O
    llama, inference, _ = import_reference()
    shape = s1mini_shape(max_seq_len=MAX_SEQ)
    w = O.random_weights(shape, seed=SEED_W, std=STD, loud=LOUD)
    out = {"std": np.float32(STD), "seed_w": np.int64(SEED_W), "n_new": np.int64(N_NEW), "max_seq_len": np.int64(MAX_SEQ),
           "loud_n": np.int64(LOUD[0]), "loud_factor": np.float32(LOUD[1]), "blocks": np.array([b[0] for b in BLOCKS])}
    prompts = {tag: make_prompt(shape, lp, seed=seed, n_vq=nvq) for tag, lp, nvq, seed in BLOCKS}
    for tag, p in prompts.items():
        out[f"{tag}.prompt"] = p.numpy()
    for prec, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        model, tok = build_reference_model(llama, shape, w, dtype)
        for tag, p in prompts.items():
            model._cache_setup_done = False
            model.max_seq_len = -1
            model.max_batch_size = -1
            seq = inference.generate(model=model, prompt=p.clone(), max_new_tokens=N_NEW, audio_masks=None,
                                     audio_parts=None, **KW)
            out[f"{tag}.{prec}.seq"] = seq.numpy().copy()
            print(tag, prec, "reference done", flush=True)
        del model
        orc = O.AROracle(shape, w, dtype)
        for tag, p in prompts.items():
            taps = []
            orc.reset()
            want = orc.generate(p.clone(), N_NEW, frame_taps=taps, **KW).numpy()
            assert np.array_equal(want, out[f"{tag}.{prec}.seq"]), f"oracle != reference at s1-mini shapes ({tag}, {prec})"
            m = record(shape, taps, out, f"{tag}.{prec}")
            print(tag, prec, out[f"{tag}.{prec}.seq"][:, -N_NEW:-N_NEW + 3], "min margin", m.min(), flush=True)
        del orc
    np.savez_compressed(os.path.join(OUT, "ar_s1mini_tf_long.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
