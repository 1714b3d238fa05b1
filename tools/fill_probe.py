"""A batch scheduler's fill at s1-mini shapes: the prompt passes of B slots as ONE ragged pass (ft_ar_prefill_slow_many)
against one pass per prompt (FT_NO_RAGGED_PREFILL=1), first frames included (one lock-step pass either way).
   python tools/fill_probe.py [B] [min_len] [max_len]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402,F401
from fish_tts_amd.ar_engine import ARHipEngine  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402
from fish_tts_amd.weights import random_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 96
args = s1_mini_args(max_seq_len=4096)
tok = ByteTokenizer()
sd = random_state_dict(args, seed=0)
eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16",
                  max_batch=B, max_new_tokens=64)
eng.load_state_dict(sd)
rng = np.random.default_rng(3)
lens = rng.integers(lo, hi + 1, B)
prompts = []
for L in lens:
    p = np.zeros((11, int(L)), dtype=np.int32)
    p[0] = rng.integers(0, tok.n_ranks, int(L))
    prompts.append(p)
sps = [eng._sampling(0.7, 0.8, 1.1, seed=i, ban_eos=True) for i in range(B)]
params = sum(v.numel() for k, v in sd.items() if k.startswith("layers."))
rows = int(lens.sum())
for name, env in (("one ragged pass", None), ("one pass per prompt", "1")):
    if env:
        os.environ["FT_NO_RAGGED_PREFILL"] = env
    else:
        os.environ.pop("FT_NO_RAGGED_PREFILL", None)
    best = 1e9
    for rep in range(4):
        eng.sync()
        t0 = time.perf_counter()
        eng.prefill_many(prompts, sps, 0)
        best = min(best, time.perf_counter() - t0)
    print(f"{B} prompts of {lo}..{hi} positions ({rows} rows), {name:20s}: {best * 1e3:8.3f} ms to every first frame "
          f"({2 * params * rows / best / 1e12:6.1f} TFLOP/s in the slow-layer products)")
eng.close()
