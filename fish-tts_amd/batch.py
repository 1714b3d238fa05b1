"""Many utterances on one GPU: a lock-step batch of slots with refill (continuous batching).

The reference serves one utterance at a time (synthesizer.py:431-481 under one model instance); BASELINE configs[2]
(batch=32 mixed-length utterances, hipGraph-captured decode loop) asks for the batched form.  Every slot is an
independent utterance (own prompt, K/V cache, position, penalty window, RNG stream keyed by its seed), all slots
advance one frame per captured graph replay, a slot that emits <|im_end|> or reaches its frame budget is retired
between bursts and the next waiting utterances are prefilled into the freed slots together; idle slots are parked.  Per utterance the result
is what a single-slot run with the same seed produces: bit for bit on an engine of up to 4 slots (and in fp16 / fp32); an
engine of >= 5 bf16 slots runs its prompt passes and frames in the MFMA batch form, whose sums take another order - those
follow the oracle within the bf16 evaluation-order margin (tests/test_ar_gpu.py)."""
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


def run_batch(engine: ARHipEngine, utterances: Sequence[Utterance], burst: int = 8, on_frames=None, on_done=None) -> dict:
    """Fills `frames` of every utterance.  `burst` = frames per scheduling step (graph replays between host looks);
    `on_frames(index, (R, k) block)` is called as blocks arrive (streaming consumers), `on_done(index)` when an
    utterance has its last frame (a codec worker can decode it while the others go on).  Returns counters:
    lock-step frame steps run and their summed widths.

    Scheduling: longest frame budget first (LPT), the longest into the lowest slots; the lock-step width of a burst is
    1 + the highest active slot, so that while the queue drains the batch narrows (cheaper frames) instead of
    dragging finished slots along."""
    B = engine.max_batch
    order = sorted(range(len(utterances)), key=lambda i: -engine._clamp_new(utterances[i].prompt.shape[1],
                                                                             utterances[i].max_new_tokens))
    waiting = deque(order)
    owner: List[Optional[int]] = [None] * B
    budget = [0] * B
    idle_sp = engine._sampling(0.7, 0.8, 1.0)
    sps = [idle_sp] * B
    parked = [False] * B

    stats = {"frame_steps": 0, "slot_frames": 0}
    # A draining batch of 2..4 rows would fall back to the multi-row GEMV frame (~325 launches, 3.07 ms at s1-mini shapes)
    # while 5 rows run the MFMA launches (2.22 ms): where the engine has that path and the batch was wide to begin with,
    # the last rows keep it by taking parked slots along.  (Batches of <= 4 slots never leave the GEMV path: their rows
    # stay bit-equal to single runs.)
    path = engine.frame_path() if hasattr(engine, "frame_path") else ""
    wide_from = 5 if "MFMA launches" in path else B + 1

    def emit(i: int, block: np.ndarray):
        if block.shape[1] == 0:
            return
        utterances[i].frames.extend(block.T)            # rows of block.T are the (R,) columns
        if on_frames is not None:
            on_frames(i, block)

    def done(i: int):
        if on_done is not None:
            on_done(i)

    def fill_many(slots) -> None:
        """The free slots take the next waiting utterances: their prompt passes in ONE call (from 5 prompts as the rows of
        one pass through the slow stack, ar_engine.prefill_many), their first frames in lock step per contiguous run of
        slots (the reference: one prompt pass per utterance, inference.py:353-362).  A slot whose utterance is over
        at its first frame takes the next one; slots nothing waits for are parked."""
        free = list(slots)
        while free and waiting:
            take = free[:len(waiting)]
            free = free[len(take):]
            ids = [waiting.popleft() for _ in take]
            us = [utterances[i] for i in ids]
            sp = [engine._sampling(u.temperature, u.top_p, u.repetition_penalty, u.seed, u.ban_eos) for u in us]
            firsts = engine.prefill_many([np.ascontiguousarray(u.prompt, dtype=np.int32) for u in us], sp, take,
                                         [u.prefix for u in us])
            for s, i, u, spi, first in zip(take, ids, us, sp, firsts):
                parked[s] = False
                emit(i, first[:, None])
                n_new = engine._clamp_new(u.prompt.shape[1], u.max_new_tokens)
                if n_new <= 1 or first[0] == engine.im_end_id:
                    done(i)
                    owner[s], budget[s], sps[s] = None, 0, idle_sp
                    free.append(s)                            # finished at its first frame: the slot takes the next one
                else:
                    owner[s], budget[s], sps[s] = i, n_new - 1, spi
            free.sort()
        for s in free:
            owner[s], budget[s], sps[s] = None, 0, idle_sp
            if not parked[s]:
                engine.park(s)
                parked[s] = True

    fill_many(range(B))                                       # longest budgets first, into the lowest slots
    while True:
        active = [s for s in range(B) if owner[s] is not None]
        if not active:
            break
        width = active[-1] + 1                            # slots above the highest active one are left out
        if 2 <= width < wide_from <= B:
            width = wide_from                             # (idle slots ride along: the wider launch form is the cheaper one)
        k = min([burst] + [budget[s] for s in active])
        frames, n = engine.decode(k, sps[:width], poll=k)
        stats["frame_steps"] += k
        stats["slot_frames"] += k * width
        freed = []
        for s in active:
            i = owner[s]
            got = int(n[s])
            emit(i, frames[s, :got].T)
            budget[s] -= got
            ended = got < k or (got > 0 and frames[s, got - 1, 0] == engine.im_end_id)
            if ended or budget[s] <= 0:
                done(i)
                freed.append(s)
        fill_many(freed)                                      # every slot this burst freed, in one refill
    return stats


def run_batch_streams(engines: Sequence[ARHipEngine], utterances: Sequence[Utterance], burst: int = 8, on_frames=None,
                      on_done=None) -> List[dict]:
    """Several lock-step batches SIDE BY SIDE on one GPU: one engine (context, HIP stream, weight copy, K/V caches) per
    batch, one host thread each.  A lock-step frame is a chain of ~290 dependent launches that leaves most of the chip
    idle, so the frames of independent batches overlap: measured at s1-mini shapes, 32 slots each -
    one batch 14 400 tok/s, two 23 200, three 28 600 (profiles/r04_multistream.txt).  Utterances are dealt longest
    budget first to the least loaded engine (an utterance with a saved K/V prefix goes to the engine that owns it); each
    engine then schedules its share with run_batch.  Callbacks receive indices into `utterances` and may be called from
    any of the threads.  Per utterance the result is what run_batch on one engine gives (its draws depend on its seed,
    frame and codebook only).  Returns run_batch's counters per engine."""
    import threading
    engines = list(engines)
    if not engines:
        raise ValueError("run_batch_streams: no engine")
    shares: List[List[int]] = [[] for _ in engines]
    loads = [0] * len(engines)
    cost = [engines[0]._clamp_new(u.prompt.shape[1], u.max_new_tokens) for u in utterances]
    for i in sorted(range(len(utterances)), key=lambda i: -cost[i]):
        pf = utterances[i].prefix
        if pf is not None:
            owners = [k for k, e in enumerate(engines) if e is pf.engine]
            if not owners:
                raise ValueError("run_batch_streams: a K/V prefix belongs to none of the engines")
            k = owners[0]
        else:
            k = min(range(len(engines)), key=lambda k: (loads[k], k))
        shares[k].append(i)
        loads[k] += cost[i]
    stats: List[Optional[dict]] = [None] * len(engines)
    errors: List[BaseException] = []

    def work(k: int) -> None:
        share = shares[k]
        try:
            stats[k] = run_batch(engines[k], [utterances[i] for i in share], burst=burst,
                                 on_frames=(lambda j, blk: on_frames(share[j], blk)) if on_frames is not None else None,
                                 on_done=(lambda j: on_done(share[j])) if on_done is not None else None)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(len(engines))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return stats
