"""Codec encode latency at the real DAC shapes (encode_reference): python tools/encode_probe.py 10 30"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402,F401
from fish_tts_amd.codec_engine import CodecHipEngine  # noqa: E402

secs = [float(x) for x in sys.argv[1:]] or [10.0, 30.0]
eng = CodecHipEngine.synthetic(device=0, max_frames=2056, seed=0, with_encoder=True)
rng = np.random.default_rng(0)
for sec in secs:
    n = int(sec * 44100)
    t = np.arange(n) / 44100.0
    audio = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        codes = eng.encode(audio)
        best = min(best, time.perf_counter() - t0)
    t0 = time.perf_counter()
    back = eng.decode(codes)
    dt = time.perf_counter() - t0
    print(f"{sec:5.1f} s of audio: encode {best * 1e3:8.2f} ms -> codes {codes.shape}, semantic range [{codes[0].min()}, {codes[0].max()}], "
          f"residual max {codes[1:].max()}; decode of those codes {dt * 1e3:7.2f} ms, finite={np.isfinite(back).all()}")
eng.close()
