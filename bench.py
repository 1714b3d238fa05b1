"""Headline benchmark: semantic tokens/sec (+ RTF) of the dual-AR decode + codec decode hot path on
synthetic 10 s utterances at the openaudio-s1-mini shapes (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one utterance through the hot path: prefill of a 48-token prompt, 215 generated frames
(10 s of audio; <|im_end|> masked because random weights never emit a meaningful EOS), then one codec
decode of the (10, 215) codes.  Weights and prompts are resident in HBM before the timed region.
With N > 1 every rank runs its own utterances (the path shards by utterance, no data-path
collective); weights are broadcast once from rank 0 over RCCL.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_FRAMES = 215         # 10 s at 21.53 frames/s
PROMPT_LEN = 48


def synth_weights(args, seed=0):
    """Random-init weights by the reference's rule (llama.py:455-464: normal(0, initializer_range) for
    Linear/Embedding, ones for norm gains), bf16."""
    from fish_tts_amd.weights import random_state_dict
    return random_state_dict(args, seed=seed, dtype=torch.bfloat16)


def synth_prompt(tok, seed=1):
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(11, PROMPT_LEN, dtype=torch.int32)
    p[0] = torch.randint(0, tok.n_ranks, (PROMPT_LEN,), generator=g)
    p[0, 0] = tok.get_token_id("<|interleave|>")
    return p.numpy()


def cpu_baseline(args_dict, sd, prompt, tok, frames=8):
    """The oracle (CPU restatement of the reference eager path, bf16) timed on the host cores on a
    bounded sample: prefill of the same prompt + `frames` decode frames."""
    from oracle import ar as O
    shape = O.ARShape(**{k: v for k, v in args_dict.items() if k in O.ARShape.__dataclass_fields__},
                      semantic_begin_id=tok.semantic_begin_id, semantic_end_id=tok.semantic_end_id,
                      im_end_id=tok.get_token_id("<|im_end|>"))
    orc = O.AROracle(shape, sd, torch.bfloat16)
    t0 = time.perf_counter()
    seq = orc.generate(torch.from_numpy(prompt), frames, temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    dt = time.perf_counter() - t0
    n = seq.shape[1] - prompt.shape[1]
    return {"value": round(n / dt, 3), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"prefill of the same {PROMPT_LEN}-token prompt + {n} greedy frames, bf16, torch CPU eager "
                      f"({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch-probe", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=3)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import fish_tts_amd  # noqa: F401
    from fish_tts_amd.ar_engine import ARHipEngine
    from fish_tts_amd.config import s1_mini_args
    from fish_tts_amd.tokenizer import ByteTokenizer

    args = s1_mini_args()
    tok = ByteTokenizer()
    prompt = synth_prompt(tok)
    im_end = tok.get_token_id("<|im_end|>")

    # weights: generated on rank 0, broadcast once over RCCL, then copied into the engine's own HBM
    sd = synth_weights(args) if rank == 0 else None
    if world > 1:
        from fish_tts_amd.parallel import broadcast_state_dict
        sd = broadcast_state_dict(sd, args, src=0, device=torch.device("cuda", local_rank))
    eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, im_end, precision="bf16",
                      device=local_rank, max_batch=1, max_new_tokens=N_FRAMES + 8)
    eng.load_state_dict(sd)
    codec = None
    try:
        from fish_tts_amd.codec_engine import CodecHipEngine
        codec = CodecHipEngine.synthetic(device=local_rank, max_frames=N_FRAMES + 8)
    except Exception as e:  # noqa: BLE001
        if rank == 0:
            print(f"[bench] codec engine unavailable: {e}", file=sys.stderr)

    kw = dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1, ban_eos=True)

    def step(i):
        t0 = time.perf_counter()
        seq = eng.generate(prompt, N_FRAMES, seed=i, poll=N_FRAMES, **kw)
        t1 = time.perf_counter()
        n = seq.shape[1] - prompt.shape[1]
        if codec is not None:
            codec.decode(seq[1:, prompt.shape[1]:][None])
        t2 = time.perf_counter()
        return n, t1 - t0, t2 - t1

    for i in range(a.warmup):
        step(1000 + i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    frames = 0
    ar_s = cod_s = 0.0
    for i in range(a.steps):
        n, ta, tc = step(i)
        frames += n
        ar_s += ta
        cod_s += tc
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        t = torch.tensor([elapsed, ar_s, cod_s, float(frames)], device="cuda", dtype=torch.float64)
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = t.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, ar_max = float(mx[0]), float(mx[1])
        frames_total = float(sm[3])
    else:
        ar_max = ar_s
        frames_total = float(frames)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # roofline of the dominant kernel family: the weight-streaming GEMV launches whose weights come from HBM
    # (4 per slow layer + the vocabulary head = 1.2 GB of the 1.303 GB algorithmic bytes of a frame-step),
    # timed live: a hipGraph of exactly those launches replayed between two HIP events on the engine's stream
    sp = eng._sampling(0.7, 0.8, 1.1, seed=7, ban_eos=True)
    ms, launches, nbytes = eng.profile_gemv(20, sp)
    achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath):  # HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes
        traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
    roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": "ft::gemv_kernel / ft::gemv_attn_combine_kernel (slow layers + vocabulary head)",
            "launches": launches, "bytes_per_launch": round(nbytes / max(launches, 1)),
            "avg_us_per_launch": round(ms * 1e3 / max(launches, 1), 3)}

    audio_s = frames_total * 2048 / 44100.0
    tok_s = frames_total / elapsed  # whole step (prefill + decode + codec) over all ranks
    out = {
        "metric": "semantic tokens/sec", "value": round(tok_s, 2),
        "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "openaudio-s1-mini shapes (BASELINE configs[1]): batch=1 per GPU, 48-token prompt, "
                               "215 frames (10 s), top-p sampling, hipGraph-captured frame step"
                               + (", + DAC codec decode" if codec is not None else ", codec decode NOT included"),
                   "frames_per_step": N_FRAMES, "prompt_len": PROMPT_LEN, "parallelism": f"replica x{world}"},
        "ar_tokens_per_s": round(frames_total / (ar_max if world > 1 else ar_s), 2),
        "rtf": round(elapsed / audio_s * world, 5) if codec is not None else None,
        "codec_ms_per_step": round(cod_s / a.steps * 1e3, 3) if codec is not None else None,
        "roofline": roof,
    }
    # extra, outside the timed region and never part of `value`: BASELINE configs[2]'s lock-step batch on this GPU
    out["lockstep_batch32_tokens_per_s"] = None
    if world == 1 and not a.no_batch_probe:
        try:
            from fish_tts_amd.config import s1_mini_args as _s1
            eng.close()
            beng = ARHipEngine(_s1(max_seq_len=1024), tok.semantic_begin_id, tok.semantic_end_id, im_end, precision="bf16",
                               device=local_rank, max_batch=32, max_new_tokens=160)
            beng.load_state_dict(sd)
            sps = [beng._sampling(0.7, 0.8, 1.1, seed=i, ban_eos=True) for i in range(32)]
            for rep in range(2):
                for b in range(32):
                    beng.prefill(prompt, sps[b], slot=b)
                beng.sync()
                t0 = time.perf_counter()
                _, nb = beng.decode(128, sps, poll=128)
                dtb = time.perf_counter() - t0
            out["lockstep_batch32_tokens_per_s"] = round(float(nb.sum()) / dtb, 1)
            beng.close()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] batch probe skipped: {e}", file=sys.stderr)
    if not a.no_cpu_baseline and world == 1:
        cpu_sd = {k: v.cpu() for k, v in sd.items()}
        out["cpu_baseline"] = cpu_baseline(args.__dict__, cpu_sd, prompt, tok, frames=a.cpu_frames)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
