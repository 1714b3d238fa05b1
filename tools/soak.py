import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import fish_tts_amd
from fish_tts_amd.ar_engine import ARHipEngine
from fish_tts_amd.batch import Utterance, run_batch
from fish_tts_amd.config import s1_mini_args
from fish_tts_amd.tokenizer import ByteTokenizer
from fish_tts_amd.weights import random_state_dict
from fish_tts_amd.codec_engine import CodecHipEngine
tok = ByteTokenizer()
args = s1_mini_args()
sd = random_state_dict(args, seed=0)
eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16", max_batch=1, max_new_tokens=2048 + 8)
eng.load_state_dict(sd)
p = np.zeros((11, 60), dtype=np.int32); p[0] = np.random.default_rng(0).integers(0, tok.n_ranks, 60)
t0 = time.perf_counter()
seq = eng.generate(p, 2048, temperature=0.7, top_p=0.8, repetition_penalty=1.1, seed=1, ban_eos=True, poll=64)
dt = time.perf_counter() - t0
n = seq.shape[1] - 60
print(f"B=1 long utterance: {n} frames in {dt:.2f} s = {n/dt:.1f} tok/s; codes in range: {(seq[1, 60:] < 4096).all() and (seq[2:, 60:] < 1024).all() and (seq[1:, 60:] >= 0).all()}")
codec = CodecHipEngine.synthetic(device=0, max_frames=2056)
t0 = time.perf_counter(); audio = codec.decode(seq[1:, 60:][None]); dt = time.perf_counter() - t0
print(f"codec decode of {n} frames ({n*2048/44100:.1f} s of audio): {dt*1e3:.1f} ms, finite={np.isfinite(audio).all()}, peak={np.abs(audio).max():.3f}")
eng.close(); codec.close()
eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16", max_batch=32, max_new_tokens=1100)
eng.load_state_dict(sd)
rng = np.random.default_rng(1)
utts = []
for i in range(48):
    q = np.zeros((11, int(rng.integers(10, 200))), dtype=np.int32); q[0] = rng.integers(0, tok.n_ranks, q.shape[1])
    utts.append(Utterance(q, int(rng.integers(50, 1000)), 0.7, 0.8, 1.1, seed=i, ban_eos=(i % 3 != 0)))
t0 = time.perf_counter(); run_batch(eng, utts, burst=16); dt = time.perf_counter() - t0
made = sum(u.columns().shape[1] for u in utts)
ok = all((u.columns()[1] < 4096).all() and (u.columns()[2:] < 1024).all() and (u.columns() >= 0).all() for u in utts)
print(f"B=32 soak: 48 utterances, {made} frames in {dt:.2f} s = {made/dt:.0f} tok/s; all codes in range: {ok}; budgets respected: {all(u.columns().shape[1] <= u.max_new_tokens for u in utts)}")
eng.close()
