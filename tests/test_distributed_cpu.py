"""The N > 1 plumbing on CPU: world_size-2 gloo processes broadcast a state dict from rank 0 and deal
utterances; no data-path collective exists (the path shards by utterance)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    import fish_tts_amd  # noqa: F401
    from fish_tts_amd.config import DualARModelArgs
    from fish_tts_amd.parallel import broadcast_state_dict, deal_utterances
    from fish_tts_amd.weights import random_state_dict
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = DualARModelArgs(vocab_size=300, n_layer=2, n_head=4, dim=64, intermediate_size=128, n_local_heads=2, head_dim=16,
                           codebook_size=64, num_codebooks=3, n_fast_layer=1, max_seq_len=64)
    sd = random_state_dict(args, seed=0, dtype=torch.float32) if rank == 0 else None
    got = broadcast_state_dict(sd, args, src=0, device=torch.device("cpu"), dtype=torch.float32)
    ref = random_state_dict(args, seed=0, dtype=torch.float32)
    ok = all(torch.equal(got[k], ref[k]) for k in ref) and set(got) == set(ref)
    mine = deal_utterances([5, 9, 3, 7], world)[rank]

    class FakeTTS:   # stands in for FishTTS(max_batch=...) on a GPU: records what this rank was asked to speak
        def synthesize_batch(self, texts, seed=0, seeds=None, **kw):
            return [f"rank{rank}:{t}:{s}".encode() for t, s in zip(texts, seeds)]
    from fish_tts_amd.parallel import synthesize_sharded
    texts = ["bb", "a", "dddd", "ccc", "eeeee"]
    wavs = synthesize_sharded(FakeTTS(), texts, dst=0, seed=100)
    sharded_ok = True
    if rank == 0:
        sharded_ok = [w.split(b":")[1].decode() for w in wavs] == texts and len({w.split(b":")[0] for w in wavs}) == world
        # utterance i was drawn with seed + i on whichever rank spoke it
        sharded_ok = sharded_ok and [int(w.split(b":")[2]) for w in wavs] == [100 + i for i in range(len(texts))]
    else:
        sharded_ok = wavs is None
    q.put((rank, ok and sharded_ok, mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_broadcast_and_deal_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(30)
    assert all(r[1] for r in res)
    assert sorted(res[0][2] + res[1][2]) == [0, 1, 2, 3]
