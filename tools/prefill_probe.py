"""Prompt-prefill latency probe at s1-mini shapes (BASELINE configs[4]: a 30 s reference is ~650 frames of prompt).
   python tools/prefill_probe.py 48 256 700 1500"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402,F401
from fish_tts_amd.ar_engine import ARHipEngine  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402
from fish_tts_amd.weights import random_state_dict  # noqa: E402

Ls = [int(x) for x in sys.argv[1:]] or [48, 256, 700, 1500]
args = s1_mini_args(max_seq_len=4096)
tok = ByteTokenizer()
sd = random_state_dict(args, seed=0)
eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16",
                  max_batch=1, max_new_tokens=64)
eng.load_state_dict(sd)
sp = eng._sampling(0.7, 0.8, 1.1, seed=0, ban_eos=True)
g = torch.Generator().manual_seed(1)
params = sum(v.numel() for k, v in sd.items() if k.startswith("layers."))
for L in Ls:
    p = torch.zeros(11, L, dtype=torch.int32)
    p[0] = torch.randint(0, tok.n_ranks, (L,), generator=g)
    best = 1e9
    for rep in range(4):
        eng.sync()
        t0 = time.perf_counter()
        eng.prefill(p.numpy(), sp)
        best = min(best, time.perf_counter() - t0)
    print(f"Lp={L:5d}: prefill {best * 1e3:8.3f} ms  ({L / best:9.0f} prompt tok/s, {2 * params * L / best / 1e12:6.1f} TFLOP/s in the slow-layer GEMMs)")
eng.close()
