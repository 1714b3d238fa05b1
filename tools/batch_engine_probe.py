"""The batch form of the codebook loop (batch_engine.h, FT_BATCH_ENGINE) against the launch path: the same 2+2-layer model at
the s1-mini widths, B lock-step rows, greedy frames; prints agreement, engine state and the frame time of both.
    python tools/batch_engine_probe.py [B=32] [frames=8]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hip_util import make_pair  # noqa: E402
from tests.shapes import make_prompt  # noqa: E402
from tests.test_ar_gpu import medium_shape  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shape = medium_shape()
prompts = [make_prompt(shape, 9 + (3 * i) % 11, seed=300 + i, n_vq=i % 4).numpy() for i in range(B)]
out = {}
for mode in ("launch", "engine"):
    os.environ.pop("FT_BATCH_ENGINE", None)
    if mode == "engine":
        os.environ["FT_BATCH_ENGINE"] = "1"
    eng, _ = make_pair(shape, "bf16", std=0.05, max_batch=B, max_new_tokens=frames + 8)
    sp = eng._sampling(0.7, 1e-6, 1.1)
    firsts = [eng.prefill(p, sp, slot=i) for i, p in enumerate(prompts)]
    t0 = time.perf_counter()
    fr, n = eng.decode(frames, [sp] * B, poll=frames)
    dt = time.perf_counter() - t0
    print(mode, "frame path:", eng.frame_path(), "| state", eng.engine_state(), f"| {dt / frames * 1e3:.3f} ms per frame (incl. graph capture)", flush=True)
    t0 = time.perf_counter()
    fr2, n2 = eng.decode(frames, [sp] * B, poll=frames)
    dt = time.perf_counter() - t0
    print(mode, f"second burst: {dt / frames * 1e3:.3f} ms per lock-step frame = {B * frames / dt:.0f} tok/s", flush=True)
    out[mode] = (np.stack(firsts), fr.copy(), n.copy())
    if mode == "engine" and os.environ.get("FT_EB_STAMPS"):
        import ctypes as C
        buf = (C.c_uint64 * (3 * 10 * 8 * 16))()
        got = eng.lib.ft_test_eb_stamps(buf, len(buf))
        nL = shape.n_fast_layer
        st = np.frombuffer(buf, dtype=np.uint64)[: 3 * shape.num_codebooks * nL * 16].reshape(3, shape.num_codebooks, nL, 16).astype(np.float64) / 100.0   # us
        names = ["x in LDS", "q k v published", "attention done", "y gathered", "Wo published", "x' gathered", "x' normalised",
                 "g published", "g gathered", "layer done"]
        for wi, wname in enumerate(("workgroup 0 (QKV role)", "workgroup 128 (Wo / head role)", "workgroup 192 (W2 role)")):
            print(wname, "- us since the previous stamp, averaged over steps 2.. and layers:")
            acc = np.zeros(10); cnt = 0
            for cb in range(2, shape.num_codebooks):
                for li in range(nL):
                    t = st[wi, cb, li, :10]
                    prev = st[wi, cb, li - 1, 9] if li > 0 else t[0]
                    d = np.diff(np.concatenate([[prev], t]))
                    acc += d; cnt += 1
            for k in range(10):
                print(f"   {names[k]:18s} {acc[k] / cnt:6.2f}")
            print(f"   {'sum':18s} {acc.sum() / cnt:6.2f}")
            tail = [st[wi, cb, nL - 1, 10:15] - st[wi, cb, nL - 1, 9] for cb in range(2, shape.num_codebooks)]
            print("   after the last layer (us since 'layer done'): head input gathered / logits published / draw done / codes known / next layer-0 x in LDS:",
                  np.round(np.mean(tail, axis=0), 2).tolist(),
                  "| step to step:", round(float(np.mean(np.diff(st[wi, 2:, 0, 0]))), 2))
    eng.close()
a, b = out["launch"], out["engine"]
print("first frames equal:", np.array_equal(a[0], b[0]))
fa, fb = a[1], b[1]
same = (fa == fb)
print(f"frames: {same.mean() * 100:.2f} % of {same.size} tokens equal; per codebook row: {[round(float(same[:, :, r].mean()), 3) for r in range(fa.shape[2])]}")
for f in range(min(frames, 4)):
    print(f" frame {f}: rows equal {int(same[:, f].all(axis=1).sum())} / {B}")
print("launch row 0:", fa[0, :2].tolist())
print("engine row 0:", fb[0, :2].tolist())
