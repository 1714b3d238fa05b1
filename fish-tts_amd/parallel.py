"""Multi-GPU plumbing.  The hot path shards by utterance (each utterance owns its prompt, KV cache,
RNG stream and codec decode), so there is no data-path collective: the only
exchange is a one-time broadcast of the weights from rank 0 over RCCL (torch.distributed 'nccl')."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], args, src: int, device: torch.device,
                         shapes: Optional[Dict[str, tuple]] = None, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Every rank returns the full state dict as tensors on `device`.  Tensors are packed into a few
    large flat buckets (one big transfer per bucket keeps all xGMI links of the root busy instead of
    paying a launch + rendezvous per small tensor)."""
    import torch.distributed as dist
    from .weights import state_dict_shapes
    shapes = shapes or state_dict_shapes(args)
    names = list(shapes)
    numels = [int(torch.Size(shapes[n]).numel()) for n in names]
    bucket_elems = 256 * 1024 * 1024  # 512 MB of bf16 per bucket
    out: Dict[str, torch.Tensor] = {}
    i = 0
    while i < len(names):
        j, tot = i, 0
        while j < len(names) and (tot == 0 or tot + numels[j] <= bucket_elems):
            tot += numels[j]
            j += 1
        flat = torch.empty(tot, dtype=dtype, device=device)
        if dist.get_rank() == src:
            off = 0
            for k in range(i, j):
                flat[off: off + numels[k]].copy_(sd[names[k]].reshape(-1).to(dtype))
                off += numels[k]
        dist.broadcast(flat, src=src)
        off = 0
        for k in range(i, j):
            out[names[k]] = flat[off: off + numels[k]].view(shapes[names[k]])
            off += numels[k]
        i = j
    return out


def deal_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time dealing of utterance indices to ranks so lock-step batches on each
    GPU have similar total length."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += lengths[i]
    return out


def synthesize_sharded(tts, texts: Sequence[str], dst: int = 0, **kw):
    """BASELINE configs[3] (a large batch across the GPUs of one node): every rank (one process per GPU, its own
    `FishTTS(max_batch=...)`) synthesises the share `deal_utterances` gives it - texts dealt by length, longest first -
    in its own lock-step batch; rank `dst` receives the WAV bytes of all texts in input order (others get None).  The
    only communication is the final gather of results; nothing crosses GPUs while decoding."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    share = deal_utterances([len(t) for t in texts], world)[rank]
    seed = kw.pop("seed", 0)
    # utterance i draws with seed + i whatever the world size and whatever else its rank was dealt
    wavs = tts.synthesize_batch([texts[i] for i in share], seeds=[seed + i for i in share], **kw) if share else []
    gathered = [None] * world if rank == dst else None
    dist.gather_object(list(zip(share, wavs)), gathered, dst=dst)
    if rank != dst:
        return None
    out = [None] * len(texts)
    for part in gathered:
        for i, w in part:
            out[i] = w
    return out
