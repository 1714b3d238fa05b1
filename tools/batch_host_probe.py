"""Where the wall clock of BASELINE configs[2] goes (32 mixed-length utterances, continuous batching): time inside
ft_ar_decode per burst length, inside the fills, and in the host code between them.   python tools/batch_host_probe.py [burst]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402,F401
import bench  # noqa: E402
from fish_tts_amd.ar_engine import ARHipEngine  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402
from fish_tts_amd.weights import random_state_dict  # noqa: E402

burst = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args = s1_mini_args(max_seq_len=4096)
tok = ByteTokenizer()
eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16",
                  max_batch=32, max_new_tokens=512)
eng.load_state_dict(random_state_dict(args, seed=0))
t_dec = collections.defaultdict(lambda: [0, 0.0])     # (k, width) -> [calls, seconds]
t_fill = [0, 0.0]
t_park = [0, 0.0]
real_decode, real_fill, real_park = eng.decode, eng.prefill_many, eng.park


def decode(k, sps, poll=8):
    t0 = time.perf_counter()
    r = real_decode(k, sps, poll)
    e = t_dec[(k, len(sps) >= 5)]
    e[0] += 1
    e[1] += time.perf_counter() - t0
    return r


def fill(*a, **kw):
    t0 = time.perf_counter()
    r = real_fill(*a, **kw)
    t_fill[0] += 1
    t_fill[1] += time.perf_counter() - t0
    return r


def park(s):
    t0 = time.perf_counter()
    real_park(s)
    t_park[0] += 1
    t_park[1] += time.perf_counter() - t0


eng.decode, eng.prefill_many, eng.park = decode, fill, park
for rep in range(2):
    t_dec.clear()
    t_fill[:] = [0, 0.0]
    t_park[:] = [0, 0.0]
    made, dt = bench.mixed_batch(eng, tok, 32, seed=2, burst=burst, reps=1)
print(f"burst {burst}: {made} frames in {dt * 1e3:.1f} ms = {made / dt:.0f} tok/s")
tot = 0.0
for (k, wide), (n, s) in sorted(t_dec.items()):
    tot += s
    print(f"  decode({k:2d} frames, {'>= 5 rows' if wide else '<= 4 rows'}): {n:3d} calls, {s * 1e3:7.1f} ms, {s / n / k * 1e3:6.3f} ms per frame")
print(f"  fills: {t_fill[0]} calls {t_fill[1] * 1e3:.1f} ms; parks: {t_park[0]} calls {t_park[1] * 1e3:.1f} ms; "
      f"decode total {tot * 1e3:.1f} ms; host code between the calls {(dt - tot - t_fill[1] - t_park[1]) * 1e3:.1f} ms")
eng.close()
