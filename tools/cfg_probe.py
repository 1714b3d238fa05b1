"""BASELINE configs[2] and [4] at s1-mini shapes on one GPU (synthetic weights, SURVEY.md §8-d cfg#3 / cfg#5).
   python tools/cfg_probe.py cfg3 | cfg5"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fish_tts_amd  # noqa: E402,F401
from fish_tts_amd.ar_engine import ARHipEngine  # noqa: E402
from fish_tts_amd.batch import Utterance, run_batch  # noqa: E402
from fish_tts_amd.config import s1_mini_args  # noqa: E402
from fish_tts_amd.tokenizer import ByteTokenizer  # noqa: E402
from fish_tts_amd.weights import random_state_dict  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
tok = ByteTokenizer()
args = s1_mini_args(max_seq_len=4096)
sd = random_state_dict(args, seed=0)


def engine(B):
    e = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id("<|im_end|>"), precision="bf16",
                    max_batch=B, max_new_tokens=512)
    e.load_state_dict(sd)
    return e


def text_prompt(rng, L):
    p = np.zeros((11, L), dtype=np.int32)
    p[0] = rng.integers(0, tok.n_ranks, L)
    return p


if which == "cfg3":
    rng = np.random.default_rng(2)
    for B, n_utt in ((32, 32), (32, 96), (8, 32)):
        eng = engine(B)
        lens = rng.integers(16, 97, n_utt)
        targets = rng.integers(108, 431, n_utt)
        for rep in range(2):
            utts = [Utterance(text_prompt(rng, int(l)), int(t), 0.7, 0.8, 1.1, seed=i, ban_eos=True)
                    for i, (l, t) in enumerate(zip(lens, targets))]
            eng.sync()
            t0 = time.perf_counter()
            run_batch(eng, utts, burst=8)
            dt = time.perf_counter() - t0
        made = sum(u.columns().shape[1] for u in utts)
        assert made == int(targets.sum()), (made, targets.sum())
        audio_s = made * 2048 / 44100
        print(f"cfg3: {n_utt} utterances (Lp U[16,96], frames U[108,430]) on {B} slots: {made} frames in {dt:.3f} s = "
              f"{made / dt:8.1f} tok/s aggregate, AR-only RTF {dt / audio_s:.5f}")
        eng.close()
else:
    rng = np.random.default_rng(3)
    B = 8
    eng = engine(B)
    ref = np.concatenate([rng.integers(0, 4096, (1, 661)), rng.integers(0, 1024, (9, 661))]).astype(np.int32)
    head = np.zeros((11, 2 + 64 + 661 + 1), dtype=np.int32)
    head[0, : 2 + 64] = rng.integers(0, tok.n_ranks, 66)
    head[0, 66: 66 + 661] = ref[0] + tok.semantic_begin_id
    head[1:, 66: 66 + 661] = ref
    head[0, -1] = tok.get_token_id("<|im_end|>")
    n_prefix = head.shape[1]
    for cached in (False, True):
        for rep in range(2):
            t_build = 0.0
            pf = None
            if cached:
                t0 = time.perf_counter()
                pf = eng.build_prefix(head)
                eng.sync()
                t_build = time.perf_counter() - t0
            utts = []
            for i in range(B):
                full = np.concatenate([head, text_prompt(rng, 49)], axis=1)
                utts.append(Utterance(full, 215, 0.7, 0.8, 1.1, seed=i, ban_eos=True, prefix=pf))
            first10 = {}
            count = [0] * B
            eng.sync()
            t0 = time.perf_counter()

            def on_frames(i, blk):
                count[i] += blk.shape[1]
                if count[i] >= 10 and i not in first10:
                    first10[i] = time.perf_counter() - t0
            run_batch(eng, utts, burst=5, on_frames=on_frames)
            dt = time.perf_counter() - t0
            if pf is not None:
                pf.free()
        made = sum(u.columns().shape[1] for u in utts)
        print(f"cfg5 ({'prefix K/V reused' if cached else 'full prefill'}): Lp={utts[0].prompt.shape[1]} (prefix {n_prefix}), "
              f"B={B}: first 10 frames of slot 0 after {first10[0] * 1e3:.1f} ms, of every slot after "
              f"{max(first10.values()) * 1e3:.1f} ms; {made} frames in {dt:.3f} s = {made / dt:.1f} tok/s"
              + (f"; prefix built once in {t_build * 1e3:.1f} ms" if cached else ""))
    eng.close()
