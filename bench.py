"""Headline benchmark: semantic tokens/sec (+ RTF) of the dual-AR decode + codec decode hot path on
synthetic 10 s utterances at the openaudio-s1-mini shapes (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W        (N > 1: this process only launches N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one utterance through the hot path: prefill of a 48-token prompt, 215 generated frames
(10 s of audio; <|im_end|> masked because random weights never emit a meaningful EOS), then one codec
decode of the (10, 215) codes.  Weights and prompts are resident in HBM before the timed region.
With N > 1 every rank runs its own utterances (the path shards by utterance, no data-path
collective); weights are broadcast once from rank 0 over RCCL.  Rank 0 prints one JSON line.

`--gpus N` without a torch.distributed environment: this process starts N fresh ranks through
`python -m torch.distributed.run` BEFORE anything here touches the GPU (a process that has initialised
the GPU is never re-executed), relays rank 0's JSON line and exits with the launcher's code.

`--config cfg4`: BASELINE configs[3] - every rank decodes 32 mixed-length utterances (prompts U[16,96], frame budgets
U[108,430], SURVEY §8-d seed 3 + rank) with continuous batching on its 32 slots; every finished utterance is decoded to
audio by a worker thread on the codec's own stream while the others go on (the pipeline of synthesizer.py:483-584 across a
batch), all of it inside the timed region; value = all ranks' frames / wall, rtf = wall / audio seconds.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_FRAMES = 215         # 10 s at 21.53 frames/s
PROMPT_LEN = 48
# algorithmic bytes of one decode frame at the s1-mini shapes: slow layers 880.8 MB + vocabulary head
# 319.0 MB + fast layers 100.7 MB (once per frame) + fast head 2.1 MB, plus 114 688 bytes of K/V per cached position
KV_BYTES_PER_POS = 114688


def frame_bytes(args):
    slow = args.n_layer * (args.dim * (args.n_head + 2 * args.n_local_heads) * args.head_dim + args.n_head * args.head_dim * args.dim
                           + 3 * args.dim * args.intermediate_size) * 2
    head = args.vocab_size * args.dim * 2
    fast = args.n_fast_layer * (args.fast_dim * (args.fast_n_head + 2 * args.fast_n_local_heads) * args.fast_head_dim
                                + args.fast_n_head * args.fast_head_dim * args.fast_dim
                                + 3 * args.fast_dim * args.fast_intermediate_size) * 2
    fhead = min(1024, args.codebook_size) * args.fast_dim * 2
    return {"slow": slow, "head": head, "fast": fast + fhead, "total": slow + head + fast + fhead}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, argv):
    """Parent of a multi-GPU run: never touches the GPU.  One child per GPU via torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif r.returncode == 0:
        print("[bench] the ranks printed no result line", file=sys.stderr)
        return 1
    return r.returncode


def synth_weights(args, seed=0):
    """Random-init weights by the reference's rule (llama.py:455-464: normal(0, initializer_range) for
    Linear/Embedding, ones for norm gains), bf16."""
    import torch
    from fish_tts_amd.weights import random_state_dict
    return random_state_dict(args, seed=seed, dtype=torch.bfloat16)


def synth_prompt(tok, seed=1, length=PROMPT_LEN):
    import torch
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(11, length, dtype=torch.int32)
    p[0] = torch.randint(0, tok.n_ranks, (length,), generator=g)
    p[0, 0] = tok.get_token_id("<|interleave|>")
    return p.numpy()


def cpu_baseline(args_dict, sd, prompt, tok, frames=32, codec_frames=N_FRAMES, threads=None):
    """the oracle (CPU restatement of the reference eager path, bf16 AR / f32 codec) timed on the host
    cores on a bounded sample of the same workload: prefill of the same prompt (timed apart), `frames` decode frames,
    one codec decode of `codec_frames` frames; RTF from those rates."""
    import torch
    from oracle import ar as O
    ncpu = os.cpu_count() or 1
    threads = threads or max(1, min(ncpu, 16))   # torch CPU eager does not scale past a socket's worth of cores here
    torch.set_num_threads(threads)
    shape = O.ARShape(**{k: v for k, v in args_dict.items() if k in O.ARShape.__dataclass_fields__},
                      semantic_begin_id=tok.semantic_begin_id, semantic_end_id=tok.semantic_end_id,
                      im_end_id=tok.get_token_id("<|im_end|>"))
    orc = O.AROracle(shape, sd, torch.bfloat16)
    kw = dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1)
    p = torch.from_numpy(prompt)
    t0 = time.perf_counter()
    orc.generate(p, 1, **kw)                       # prefill + the frame it yields
    t_pf = time.perf_counter() - t0
    t0 = time.perf_counter()
    seq = orc.generate(p, 1 + frames, **kw)
    t_all = time.perf_counter() - t0
    n = seq.shape[1] - prompt.shape[1] - 1
    dec_s = max(t_all - t_pf, 1e-9)
    out = {"value": round(n / dec_s, 3), "unit": "tokens/s", "cores": threads, "kind": "port",
           "sample": f"{n} greedy decode frames after a {prompt.shape[1]}-token prefill (prefill {t_pf:.1f} s timed apart, "
                     f"decode {dec_s:.1f} s), bf16, torch CPU eager, {threads} of {ncpu} host threads",
           "prefill_s": round(t_pf, 2)}
    try:
        from oracle import codec as OC
        cs = OC.CodecShape()
        orc_c = OC.CodecOracle(cs, OC.random_weights(cs, seed=0))
        codes = torch.randint(0, 1024, (1, cs.n_codebooks + 1, codec_frames))
        t0 = time.perf_counter()
        with torch.no_grad():
            orc_c.decode(codes, torch.tensor([codec_frames]))
        t_c = time.perf_counter() - t0
        audio_s = codec_frames * 2048 / 44100.0
        out["codec_s"] = round(t_c, 2)
        out["rtf"] = round((audio_s * 21.533 / max(n / dec_s, 1e-9) + t_c + t_pf) / audio_s, 3)
        out["sample"] += f"; one {codec_frames}-frame codec decode in f32 ({t_c:.1f} s)"
    except Exception as e:  # noqa: BLE001
        out["sample"] += f"; codec leg skipped ({type(e).__name__}: {e})"
    return out


def mixed_batch(eng, tok, n_utt, seed, burst=8, reps=2, codec=None, info=None):
    """BASELINE configs[2]: n_utt utterances (prompts U[16,96], frame budgets U[108,430]) with continuous batching;
    returns (frames, seconds) of the last of `reps` passes (the first pass warms the graphs of every batch width).
    codec: every finished utterance's codes are decoded to audio by a worker thread (its own HIP stream) while the batch
    goes on; the pass ends when the last waveform is there.  info: receives the pass's counters (lock-step frame steps,
    cached positions read, audio samples)."""
    import queue
    import threading

    import numpy as np
    from fish_tts_amd.batch import Utterance, run_batch
    rng = np.random.default_rng(seed)
    lens = rng.integers(16, 97, n_utt)
    targets = rng.integers(108, 431, n_utt)

    def prompt(L):
        p = np.zeros((11, L), dtype=np.int32)
        p[0] = rng.integers(0, tok.n_ranks, L)
        return p
    dt = 0.0
    for rep in range(reps):
        utts = [Utterance(prompt(int(l)), int(t), 0.7, 0.8, 1.1, seed=i, ban_eos=True) for i, (l, t) in enumerate(zip(lens, targets))]
        q, err, samples = queue.Queue(), [], [0]
        worker = None
        if codec is not None:
            def decode_worker():
                try:
                    while True:
                        i = q.get()
                        if i is None:
                            return
                        codes = utts[i].codes()
                        if codes.shape[1]:
                            samples[0] += codec.decode(codes[None]).shape[1]
                except Exception as e:  # noqa: BLE001
                    err.append(e)
            worker = threading.Thread(target=decode_worker, daemon=True)
            worker.start()
        eng.sync()
        t0 = time.perf_counter()
        stats = run_batch(eng, utts, burst=burst, on_done=(q.put if codec is not None else None))
        if worker is not None:
            q.put(None)
            worker.join()
            if err:
                raise err[0]
        dt = time.perf_counter() - t0
        if info is not None:
            info.update(stats)
            # cached positions the frames of this pass read: frame f of an utterance attends over Lp + f positions
            info["kv_positions"] = int(sum(sum(range(int(l), int(l) + u.columns().shape[1])) for l, u in zip(lens, utts)))
            info["audio_samples"] = samples[0]
    made = sum(u.columns().shape[1] for u in utts)
    return made, dt


def fill_probe(eng, tok, B):
    """A scheduler's fill: the prompt passes of B slots as one ragged pass (ft_ar_prefill_slow_many) and, with
    FT_NO_RAGGED_PREFILL, one pass per prompt as the reference runs them (inference.py:353-362); first frames included."""
    import numpy as np
    rng = np.random.default_rng(3)
    lens = rng.integers(16, 97, B)
    prompts = []
    for L in lens:
        p = np.zeros((11, int(L)), dtype=np.int32)
        p[0] = rng.integers(0, tok.n_ranks, int(L))
        prompts.append(p)
    sps = [eng._sampling(0.7, 0.8, 1.1, seed=i, ban_eos=True) for i in range(B)]
    res = {"prompt_positions": int(lens.sum())}
    for name, off in (("one_ragged_pass_ms", False), ("one_pass_per_prompt_ms", True)):
        if off:
            os.environ["FT_NO_RAGGED_PREFILL"] = "1"
        best = 1e9
        try:
            for rep in range(3):
                eng.sync()
                t0 = time.perf_counter()
                eng.prefill_many(prompts, sps, 0)
                best = min(best, time.perf_counter() - t0)
        finally:
            os.environ.pop("FT_NO_RAGGED_PREFILL", None)
        res[name] = round(best * 1e3, 3)
    return res


def lockstep_streams_probe(engines, tok, B, n_frames=128, prompt_len=48):
    """Several engines (one context + stream + weight copy each) decode B equal-length utterances each in lock step at the
    same time, one host thread per engine: (frames of all, seconds) of the decode loops alone."""
    import threading

    import numpy as np
    rng = np.random.default_rng(5)
    prompts = []
    for i in range(B):
        p = np.zeros((11, prompt_len), dtype=np.int32)
        p[0] = rng.integers(0, tok.n_ranks, prompt_len)
        prompts.append(p)
    sps = [[e._sampling(0.7, 0.8, 1.1, seed=1000 * k + i, ban_eos=True) for i in range(B)] for k, e in enumerate(engines)]
    dt, made = 0.0, [0] * len(engines)
    for rep in range(2):                      # the first pass captures the graphs
        for e, sp in zip(engines, sps):
            e.prefill_many(prompts, sp, 0)
            e.sync()

        def work(k):
            _, n = engines[k].decode(n_frames, sps[k], poll=n_frames)
            made[k] = int(n.sum())
        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(engines))]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
    return sum(made), dt


def lockstep_probe(eng, tok, B, n_frames=128, prompt_len=48):
    """B utterances of equal length decoded in lock step (no refill): (frames, seconds) of the decode loop alone."""
    import numpy as np
    rng = np.random.default_rng(5)
    sps = [eng._sampling(0.7, 0.8, 1.1, seed=i, ban_eos=True) for i in range(B)]
    prompts = []
    for i in range(B):
        p = np.zeros((11, prompt_len), dtype=np.int32)
        p[0] = rng.integers(0, tok.n_ranks, prompt_len)
        prompts.append(p)
    dt = 0.0
    for rep in range(2):                      # the first pass captures the graph of this width
        eng.prefill_many(prompts, sps, 0)
        eng.sync()
        t0 = time.perf_counter()
        _, n = eng.decode(n_frames, sps, poll=n_frames)
        dt = time.perf_counter() - t0
    return int(n.sum()), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg4"],
                    help="cfg2 = BASELINE configs[1] (the headline: batch 1, 10 s); cfg4 = configs[3] (32 mixed-length slots per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch-probe", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=32)
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU only: the ranks form a gloo group and rank 0 prints how many it saw (tests/test_bench_launcher.py)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    if a.launcher_selftest:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"metric": "launcher selftest", "n_gpus": int(t.item()), "requested": a.gpus}))
        dist.destroy_process_group()
        return

    import numpy as np  # noqa: F401
    import torch
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
    ranks_seen = 1
    if world > 1:
        from fish_tts_amd.parallel import broadcast_state_dict
        sd = broadcast_state_dict(sd, args, src=0, device=torch.device("cuda", local_rank))
        seen = torch.ones(1, device="cuda")
        dist.all_reduce(seen)            # n_gpus in the result line = ranks the RCCL group really has
        ranks_seen = int(seen.item())

    if a.config == "cfg4":
        from fish_tts_amd.codec_engine import CodecHipEngine
        margs = s1_mini_args(max_seq_len=4096)
        eng = ARHipEngine(margs, tok.semantic_begin_id, tok.semantic_end_id, im_end, precision="bf16", device=local_rank,
                          max_batch=32, max_new_tokens=512)
        eng.load_state_dict(sd)
        codec = CodecHipEngine.synthetic(device=local_rank, max_frames=440)
        made, steps_run, kv_pos, audio = 0, 0, 0, 0
        for i in range(max(a.warmup, 1)):     # at least one pass: it captures the graphs of every batch width
            mixed_batch(eng, tok, 32, seed=1000 + rank, reps=1, codec=codec)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        for i in range(a.steps):
            info = {}
            m, _ = mixed_batch(eng, tok, 32, seed=3 + rank + 100 * i, reps=1, codec=codec, info=info)
            made += m
            steps_run += info["frame_steps"]
            kv_pos += info["kv_positions"]
            audio += info["audio_samples"]
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t_begin
        if dist is not None:
            t = torch.tensor([elapsed, float(made), float(steps_run), float(kv_pos), float(audio)], device="cuda", dtype=torch.float64)
            mx, sm = t.clone(), t.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            elapsed, made, steps_run, kv_pos, audio = float(mx[0]), float(sm[1]), float(sm[2]), float(sm[3]), float(sm[4])
        if rank == 0:
            # roofline of the lock-step frame (the step's dominant chain): every frame step streams the 1.303 GB of weights
            # once for all its slots, every slot frame reads its own cached positions; per GPU, over the timed region
            # (prompt passes and codec decodes included in the time, not in the bytes)
            fb = frame_bytes(margs)
            nbytes = steps_run * fb["total"] + kv_pos * KV_BYTES_PER_POS
            achieved = nbytes / world / elapsed / 1e9
            audio_s = audio / 44100.0
            print(json.dumps({
                "metric": "semantic tokens/sec", "value": round(made / elapsed, 2), "unit": "tokens/s", "n_gpus": ranks_seen,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "openaudio-s1-mini shapes (BASELINE configs[3]): 32 mixed-length utterances per GPU "
                                       "(prompts U[16,96], 108-430 frames), continuous batching on 32 lock-step slots, every "
                                       "finished utterance decoded by the DAC codec on a second stream",
                           "parallelism": f"replica x{world}", "frames_total": int(made), "audio_s": round(audio_s, 2)},
                "rtf": round(elapsed / audio_s, 5) if audio_s > 0 else None,
                "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                             "traffic_source": "none: no counter pass exists for the batch workload",
                             "kernel": "lock-step decode frames (one pass over the weights per frame step for all slots + each slot's K/V), per GPU",
                             "frame_steps": int(steps_run), "ms_per_frame_step": round(elapsed * world / max(steps_run, 1) * 1e3, 4)},
                "cpu_baseline": None}))
        codec.close()
        eng.close()
        if dist is not None:
            dist.destroy_process_group()
        return

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

    # ---- roofline of the step's dominant kernel chain: ONE decode frame (slow stack, vocabulary head, semantic draw, fast
    # codebook loop), timed live with HIP events on the engine's stream.  achieved = algorithmic bytes of a frame
    # (1.303 GB of weights + 114 688 B of K/V per cached position) / measured frame time.
    sp = eng._sampling(0.7, 0.8, 1.1, seed=7, ban_eos=True)
    fb = frame_bytes(args)
    if os.environ.get("FT_NO_GRAPH"):
        # counter runs (rocprofv3 --pmc, profiles/README.md) launch every frame eagerly; the profiler does not survive
        # a stream capture, and the roofline block measures the captured graph: leave it out of such a run
        print(json.dumps({"metric": "semantic tokens/sec", "value": round(frames_total / elapsed, 2), "unit": "tokens/s",
                          "n_gpus": ranks_seen, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "openaudio-s1-mini shapes (BASELINE configs[1]), FT_NO_GRAPH: eager frames (counter runs)",
                                     "frames_per_step": N_FRAMES, "prompt_len": PROMPT_LEN, "parallelism": f"replica x{world}"},
                          "roofline": None, "cpu_baseline": None}))
        if dist is not None:
            dist.destroy_process_group()
        return
    eng.prefill(prompt, sp)
    NPF = 48
    ms_graph, seg, nodes = eng.profile_frame(NPF, sp)
    flags = eng.engine_state()[0]
    pos_mid = PROMPT_LEN + NPF            # mean cached positions over the profiled frames (about)
    per_frame_bytes = fb["total"] + KV_BYTES_PER_POS * pos_mid
    frame_ms = ms_graph / NPF
    achieved = per_frame_bytes / (frame_ms * 1e-3) / 1e9
    # HBM-side bytes per frame: NOT measured by this run - the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    # over this bench with eager frames (FT_NO_GRAPH; counters and stream capture do not go together), profiles/README.md
    traffic, traffic_source = None, None
    for rr in ("r04", "r03", "r02"):
        tpath = os.path.join(ROOT, "profiles", rr + "_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_frame")
            traffic_source = f"profiles/{rr}_traffic.json (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes, eager frames; not this run)"
            break
    parts = {}
    for name, ms_k, nbytes in (("slow_stack", seg[0], fb["slow"] + KV_BYTES_PER_POS * pos_mid), ("head_and_draw", seg[1], fb["head"]),
                               ("fast_loop", seg[2], fb["fast"])):
        per = ms_k / max(NPF - 1, 1)
        parts[name] = {"ms": round(per, 4), "algorithmic_GB": round(nbytes / 1e9, 4),
                       "GBps": round(nbytes / (per * 1e-3) / 1e9, 1) if per > 0 else None}
    roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": ("one decode frame = ft::slow_engine_kernel + vocabulary head gemv + semantic draw + ft::fast_engine_kernel"
                       if flags == 3 else "one decode frame (launch path: gemv / attention / sampler kernels)"),
            "launches_per_frame": nodes, "bytes_per_frame": int(per_frame_bytes), "frame_us": round(frame_ms * 1e3, 2),
            "frame_ms": round(frame_ms, 4), "frame_path": eng.frame_path(), "parts": parts}
    try:   # sub-field kept from round 1: the HBM-streamed GEMV launches of the launch path alone
        ms, launches, nbytes = eng.profile_gemv(20, sp)
        roof["gemv_launch_path"] = {"GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "launches": launches,
                                    "avg_us_per_launch": round(ms * 1e3 / max(launches, 1), 3)}
    except Exception as e:  # noqa: BLE001
        print(f"[bench] gemv sub-profile skipped: {e}", file=sys.stderr)

    audio_s = frames_total * 2048 / 44100.0
    tok_s = frames_total / elapsed  # whole step (prefill + decode + codec) over all ranks
    out = {
        "metric": "semantic tokens/sec", "value": round(tok_s, 2),
        "unit": "tokens/s", "n_gpus": ranks_seen, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "openaudio-s1-mini shapes (BASELINE configs[1]): batch=1 per GPU, 48-token prompt, "
                               "215 frames (10 s), top-p sampling, hipGraph-captured frame step"
                               + (", + DAC codec decode" if codec is not None else ", codec decode NOT included"),
                   "frames_per_step": N_FRAMES, "prompt_len": PROMPT_LEN, "parallelism": f"replica x{world}",
                   "frame_engine": flags},
        "ar_tokens_per_s": round(frames_total / (ar_max if world > 1 else ar_s), 2),
        "rtf": round(elapsed / audio_s * world, 5) if codec is not None else None,
        "codec_ms_per_step": round(cod_s / a.steps * 1e3, 3) if codec is not None else None,
        "roofline": roof,
    }
    # ---- extras, outside the timed region and never part of `value`, each with its own wall clock and frame count:
    # BASELINE configs[2] (32 slots, mixed lengths, continuous batching) and configs[4] (voice cloning, B = 8, streamed)
    out["configs"] = {}
    if world == 1 and not a.no_batch_probe:
        try:
            eng.close()
            margs = s1_mini_args(max_seq_len=4096)
            beng = ARHipEngine(margs, tok.semantic_begin_id, tok.semantic_end_id, im_end, precision="bf16",
                               device=local_rank, max_batch=32, max_new_tokens=512)
            beng.load_state_dict(sd)
            made, dtm = mixed_batch(beng, tok, 32, seed=2)
            out["configs"]["configs[2] batch=32 mixed lengths, continuous batching"] = {
                "frames": int(made), "wall_s": round(dtm, 4), "tokens_per_s": round(made / dtm, 1),
                "ar_rtf": round(dtm / (made * 2048 / 44100.0), 5)}
            out["configs"]["fill of 32 slots, prompts U[16,96] (prompt passes + first frames)"] = fill_probe(beng, tok, 32)
            made, dtm = lockstep_probe(beng, tok, 32)
            out["configs"]["lock-step B=32, 48-token prompts, 128 frames (decode loop alone)"] = {
                "frames": int(made), "wall_s": round(dtm, 4), "tokens_per_s": round(made / dtm, 1),
                "ms_per_lockstep_frame": round(dtm / 128 * 1e3, 4)}
            # beyond BASELINE's batch sizes: lock-step batches side by side (a frame is a chain of dependent launches that
            # leaves most of the chip idle, so independent batches overlap): 3 engines x 32 slots = 96 utterances on the GPU
            more = []
            try:
                for _ in range(2):
                    e2 = ARHipEngine(margs, tok.semantic_begin_id, tok.semantic_end_id, im_end, precision="bf16",
                                     device=local_rank, max_batch=32, max_new_tokens=512)
                    more.append(e2)
                    e2.load_state_dict(sd)
                for n_eng in (2, 3):
                    made, dtm = lockstep_streams_probe([beng] + more[: n_eng - 1], tok, 32)
                    out["configs"][f"{n_eng} lock-step batches of 32 side by side ({32 * n_eng} utterances, one stream each), 128 frames"] = {
                        "frames": int(made), "wall_s": round(dtm, 4), "tokens_per_s": round(made / dtm, 1),
                        "ms_per_lockstep_frame_each": round(dtm / 128 * 1e3, 4)}
            finally:
                for e2 in more:
                    e2.close()
            beng.close()
            out["configs"]["configs[4] voice cloning B=8 streamed"] = voice_cloning_probe(sd, tok, im_end, local_rank)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] batch probes skipped: {type(e).__name__}: {e}", file=sys.stderr)
    if not a.no_cpu_baseline and world == 1:
        cpu_sd = {k: v.cpu() for k, v in sd.items()}
        out["cpu_baseline"] = cpu_baseline(args.__dict__, cpu_sd, prompt, tok, frames=a.cpu_frames)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def voice_cloning_probe(sd, tok, im_end, device):
    """BASELINE configs[4]: 8 utterances that share a 30 s reference (661 frames) + 49 text tokens each, reference K/V
    built once and restored per slot, frames streamed in bursts of 5; time to the 10th frame of every slot + tok/s."""
    import numpy as np
    from fish_tts_amd.ar_engine import ARHipEngine
    from fish_tts_amd.batch import Utterance, run_batch
    from fish_tts_amd.config import s1_mini_args
    rng = np.random.default_rng(3)
    B = 8
    eng = ARHipEngine(s1_mini_args(max_seq_len=4096), tok.semantic_begin_id, tok.semantic_end_id, im_end, precision="bf16",
                      device=device, max_batch=B, max_new_tokens=512)
    eng.load_state_dict(sd)
    ref = np.concatenate([rng.integers(0, 4096, (1, 661)), rng.integers(0, 1024, (9, 661))]).astype(np.int32)
    head = np.zeros((11, 2 + 64 + 661 + 1), dtype=np.int32)
    head[0, : 2 + 64] = rng.integers(0, tok.n_ranks, 66)
    head[0, 66: 66 + 661] = ref[0] + tok.semantic_begin_id
    head[1:, 66: 66 + 661] = ref
    head[0, -1] = im_end
    res = {}
    for rep in range(2):
        t0 = time.perf_counter()
        pf = eng.build_prefix(head)
        eng.sync()
        t_build = time.perf_counter() - t0
        utts = []
        for i in range(B):
            text = np.zeros((11, 49), dtype=np.int32)
            text[0] = rng.integers(0, tok.n_ranks, 49)
            utts.append(Utterance(np.concatenate([head, text], axis=1), 215, 0.7, 0.8, 1.1, seed=i, ban_eos=True, prefix=pf))
        first10, count = {}, [0] * B
        eng.sync()
        t0 = time.perf_counter()

        def on_frames(i, blk):
            count[i] += blk.shape[1]
            if count[i] >= 10 and i not in first10:
                first10[i] = time.perf_counter() - t0
        run_batch(eng, utts, burst=5, on_frames=on_frames)
        dt = time.perf_counter() - t0
        pf.free()
        made = sum(u.columns().shape[1] for u in utts)
        res = {"frames": int(made), "wall_s": round(dt, 4), "tokens_per_s": round(made / dt, 1),
               "first_10_frames_all_slots_ms": round(max(first10.values()) * 1e3, 1), "reference_kv_build_ms": round(t_build * 1e3, 1),
               "prompt_len": int(utts[0].prompt.shape[1])}
    eng.close()
    return res


if __name__ == "__main__":
    main()
