"""Decode speed against context length at B=1 (s1-mini shapes): python tools/longctx_probe.py 48 800 2000 4000"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402,F401
from fish_tts_amd.ar_engine import ARHipEngine  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402
from fish_tts_amd.weights import random_state_dict  # noqa: E402

Ls = [int(x) for x in sys.argv[1:]] or [48, 800, 2000, 4000]
args = s1_mini_args(max_seq_len=8192)
tok = ByteTokenizer()
eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16",
                  max_batch=1, max_new_tokens=160)
eng.load_state_dict(random_state_dict(args, seed=0))
sp = eng._sampling(0.7, 0.8, 1.1, seed=0, ban_eos=True)
g = torch.Generator().manual_seed(1)
for L in Ls:
    p = torch.zeros(11, L, dtype=torch.int32)
    p[0] = torch.randint(0, tok.n_ranks, (L,), generator=g)
    for rep in range(2):
        eng.prefill(p.numpy(), sp)
        eng.sync()
        t0 = time.perf_counter()
        fr, n = eng.decode(128, [sp], poll=128)
        dt = time.perf_counter() - t0
    print(f"context {L:5d}..{L + 128:5d}: {128 / dt:7.1f} tok/s, {dt / 128 * 1e3:6.3f} ms per frame (nsplit={os.environ.get('FT_ATTN_NSPLIT', 'default')})")
eng.close()
