"""Product-conditions soak of the persistent frame engine at the FULL s1-mini depth (28 + 4 layers, V = 155 776, bf16) with
the codec at its real widths: `synthesize_stream` (decoder thread decoding chunks beside the AR loop) and `synthesize`
alternate for SECONDS seconds; a second thread decodes 215-frame utterances on another codec context all the while.
Reports utterances, frames, audio seconds, wall clock, and the engine's state (time-outs recovered must stay 0).
    python tools/soak_stream.py [SECONDS=90]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import fish_tts_amd as ft  # noqa: E402
from fish_tts_amd.codec_engine import CodecHipEngine  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
tok = ByteTokenizer()
synth = ft.FishTTS.synthetic(s1_mini_args(), tok, precision="bf16", seed=0, warmup=True, max_new_tokens=1024)
print("frame path:", synth._engine.frame_path(), flush=True)
flags0 = synth._engine.engine_state()[0]
stop = threading.Event()
side = {"n": 0}


def side_decoder():
    codec = CodecHipEngine.synthetic(device=0, max_frames=224, seed=1)
    g = np.random.default_rng(3)
    codes = np.zeros((1, 10, 215), dtype=np.int32)
    codes[:, 0] = g.integers(0, 4096, (1, 215)); codes[:, 1:] = g.integers(0, 1024, (1, 9, 215))
    while not stop.is_set():
        codec.decode(codes)
        side["n"] += 1
    codec.close()


th = threading.Thread(target=side_decoder)
th.start()
texts = ["The quick brown fox jumps over the lazy dog.", "Streaming and batch share one engine.",
         "A considerably longer sentence, so that the prompt pass and the number of frames change from call to call."]
t0 = time.perf_counter()
n_utt = n_pcm = 0
last = t0
while time.perf_counter() - t0 < secs:
    text = texts[n_utt % 3]
    budget = 100 + 60 * (n_utt % 5)
    if n_utt % 2 == 0:
        for pcm in synth.synthesize_stream(text, chunk_tokens=20, min_first_chunk=10, max_tokens=budget, temperature=0.7, top_p=0.8,
                                           repetition_penalty=1.1):
            n_pcm += len(pcm) // 2
    else:
        wav = synth.synthesize(text, max_tokens=budget, temperature=0.7, top_p=0.8, repetition_penalty=1.1)
        n_pcm += (len(wav) - 44) // 2
    n_utt += 1
    if time.perf_counter() - last > 20:
        last = time.perf_counter()
        print(f"  {last - t0:6.1f} s: {n_utt} utterances, engine state {synth._engine.engine_state()}", flush=True)
dt = time.perf_counter() - t0
stop.set()
th.join()
flags, aborted, where = synth._engine.engine_state()
print(f"{n_utt} utterances ({n_utt // 2 + n_utt % 2} streamed), {n_pcm / 44100:.1f} s of audio in {dt:.1f} s wall (RTF {dt / (n_pcm / 44100):.4f} with a second "
      f"codec context decoding {side['n']} x 215 frames beside it)")
print(f"engine flags {flags} (at start {flags0}), hand-off time-outs recovered: {aborted}, last where {where}")
print("frame path:", synth._engine.frame_path())
synth._engine.close()
synth._vocoder.close()
sys.exit(0 if (aborted == 0 and flags == flags0 == 3) else 1)
