"""Independent B=1 decode graphs on separate HIP streams of one GPU (one engine = one context + stream + weight copy),
driven from one thread each: does latency-bound single-utterance decode overlap?  python tools/multistream_probe.py 1 2 4 8"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402,F401
from fish_tts_amd.ar_engine import ARHipEngine  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402
from fish_tts_amd.weights import random_state_dict  # noqa: E402

# arguments: N (streams of one utterance each) or NxB (N streams, each a lock-step batch of B)
specs = [tuple(int(v) for v in (x.split("x") + ["1"])[:2]) for x in (sys.argv[1:] or ["1", "2", "4"])]
args = s1_mini_args(max_seq_len=1024)
tok = ByteTokenizer()
sd = random_state_dict(args, seed=0)
frames = 128
rng = np.random.default_rng(0)
prompt = np.zeros((11, 48), dtype=np.int32)
prompt[0] = rng.integers(0, tok.n_ranks, 48)
for N, B in specs:
    engines = []
    for _ in range(N):
        e = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16",
                        max_batch=B, max_new_tokens=frames + 8)
        e.load_state_dict(sd)
        engines.append(e)
    for rep in range(2):
        sps = [[e._sampling(0.7, 0.8, 1.1, seed=100 * i + b, ban_eos=True) for b in range(B)] for i, e in enumerate(engines)]
        for e, sp in zip(engines, sps):
            for b in range(B):
                e.prefill(prompt, sp[b], slot=b)
            e.sync()
        out = [0] * N

        def work(i):
            _, n = engines[i].decode(frames, sps[i], poll=frames)
            out[i] = int(n.sum())
        ths = [threading.Thread(target=work, args=(i,)) for i in range(N)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
    print(f"{N} independent streams x lock-step batch {B}: {sum(out) / dt:8.1f} tok/s aggregate ({dt / frames * 1e3:.3f} ms per frame each)")
    for e in engines:
        e.close()
