"""Lock-step batch scaling probe (BASELINE configs[2]): B utterances decoded in one captured graph.
   python tools/batch_probe.py 1 2 4 8 16"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402
from fish_tts_amd.ar_engine import ARHipEngine  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402
from fish_tts_amd.weights import random_state_dict  # noqa: E402

Bs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
args = s1_mini_args(max_seq_len=1024)
tok = ByteTokenizer()
sd = random_state_dict(args, seed=0)
frames = 128
for B in Bs:
    eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16",
                      max_batch=B, max_new_tokens=frames + 8)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    sps = [eng._sampling(0.7, 0.8, 1.1, seed=i, ban_eos=True) for i in range(B)]
    for rep in range(2):
        for b in range(B):
            p = torch.zeros(11, 32 + 8 * (b % 4), dtype=torch.int32)
            p[0] = torch.randint(0, tok.n_ranks, (p.shape[1],), generator=g)
            eng.prefill(p.numpy(), sps[b], slot=b)
        eng.sync()
        t0 = time.perf_counter()
        fr, n = eng.decode(frames, sps, poll=frames)
        dt = time.perf_counter() - t0
    print(f"B={B:3d}: {frames * B / dt:9.1f} tok/s aggregate, {dt / frames * 1e3:7.3f} ms per lock-step frame, n={n.tolist()[:4]}")
    eng.close()
