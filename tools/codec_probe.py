"""Times one codec decode of T frames (synthetic weights) a few times: python tools/codec_probe.py 215"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa
from fish_tts_amd.codec_engine import CodecHipEngine
T = int(sys.argv[1]) if len(sys.argv) > 1 else 215
eng = CodecHipEngine.synthetic(max_frames=T + 8)
g = np.random.default_rng(0)
codes = np.zeros((1, 10, T), dtype=np.int32)
codes[:, 0] = g.integers(0, 4096, (1, T)); codes[:, 1:] = g.integers(0, 1024, (1, 9, T))
for i in range(4):
    t0 = time.perf_counter(); a = eng.decode(codes); dt = time.perf_counter() - t0
    print(f"decode {T} frames: {dt*1e3:.2f} ms  ({6.76*T/dt/1e3:.1f} TFLOP/s)  rms={float(np.sqrt((a**2).mean())):.3f}")
