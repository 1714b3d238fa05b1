"""Many utterances on one GPU: a lock-step batch of slots with refill (continuous batching).

The reference serves one utterance at a time (synthesizer.py:431-481 under one model instance); BASELINE configs[2]
(batch=32 mixed-length utterances, hipGraph-captured decode loop) asks for the batched form.  Every slot is an
independent utterance (own prompt, K/V cache, position, penalty window, RNG stream keyed by its seed), all slots
advance one frame per captured graph replay, a slot that emits <|im_end|> or reaches its frame budget is retired
between bursts and the next waiting utterance is prefilled into it; idle slots are parked.  Per utterance the result
is what a single-slot run with the same seed produces (tests/test_ar_gpu.py)."""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from .ar_engine import ARHipEngine, KVPrefix


@dataclass
class Utterance:
    prompt: np.ndarray                    # (num_codebooks + 1, Lp) int32
    max_new_tokens: int = 2048
    temperature: float = 0.7
    top_p: float = 0.8
    repetition_penalty: float = 1.1
    seed: int = 0
    ban_eos: bool = False
    prefix: Optional[KVPrefix] = None     # saved K/V of the first prefix.n_pos prompt columns
    frames: List[np.ndarray] = field(default_factory=list)   # (R,) generated columns, <|im_end|> column included

    def columns(self) -> np.ndarray:
        """(R, n) generated columns."""
        R = self.prompt.shape[0]
        return np.stack(self.frames, axis=1) if self.frames else np.zeros((R, 0), dtype=np.int32)

    def codes(self) -> np.ndarray:
        """(num_codebooks, n - 1): batch-mode result, the last generated column dropped (inference.py:839)."""
        return self.columns()[1:, :-1].copy()


def run_batch(engine: ARHipEngine, utterances: Sequence[Utterance], burst: int = 8, on_frames=None) -> None:
    """Fills `frames` of every utterance.  `burst` = frames per scheduling step (graph replays between host looks);
    `on_frames(index, (R, k) block)` is called as blocks arrive (streaming consumers)."""
    B = engine.max_batch
    waiting = deque(range(len(utterances)))
    owner: List[Optional[int]] = [None] * B
    budget = [0] * B
    idle_sp = engine._sampling(0.7, 0.8, 1.0)
    sps = [idle_sp] * B

    def emit(i: int, block: np.ndarray):
        u = utterances[i]
        u.frames.extend(block[:, j] for j in range(block.shape[1]))
        if on_frames is not None and block.shape[1]:
            on_frames(i, block)

    def fill(slot: int) -> None:
        while waiting:
            i = waiting.popleft()
            u = utterances[i]
            n_new = engine._clamp_new(u.prompt.shape[1], u.max_new_tokens)
            sp = engine._sampling(u.temperature, u.top_p, u.repetition_penalty, u.seed, u.ban_eos)
            first = engine._start(np.ascontiguousarray(u.prompt, dtype=np.int32), sp, u.prefix, slot)
            emit(i, first[:, None])
            if n_new <= 1 or first[0] == engine.im_end_id:
                continue                                  # finished at its first frame: the slot takes the next one
            owner[slot], budget[slot], sps[slot] = i, n_new - 1, sp
            return
        owner[slot], budget[slot], sps[slot] = None, 0, idle_sp
        engine.park(slot)

    for s in range(B):
        fill(s)
    while any(o is not None for o in owner):
        k = min([burst] + [budget[s] for s in range(B) if owner[s] is not None])
        frames, n = engine.decode(k, sps, poll=k)
        for s in range(B):
            i = owner[s]
            if i is None:
                continue
            got = int(n[s])
            emit(i, frames[s, :got].T)
            budget[s] -= got
            ended = got < k or (got > 0 and frames[s, got - 1, 0] == engine.im_end_id)
            if ended or budget[s] <= 0:
                fill(s)
