"""`python bench.py --gpus N` with no torch.distributed environment must start N ranks itself (the form the driver
uses for the multi-GPU scaling runs) and report the number of ranks the process group really has.  Here on the CPU:
the launcher path with 2 gloo ranks (--launcher-selftest touches no GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_bench_gpus_flag_launches_that_many_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["requested"] == 2


def test_bench_single_rank_needs_no_launcher():
    """--gpus 1 (the default) runs in-process: no child launcher, no process group."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from fish_tts_amd.config import s1_mini_args
    fb = m.frame_bytes(s1_mini_args())
    # 880.8 MB + 319.0 MB + 100.7 MB + 2.1 MB = 1.303 GB per frame-step
    assert round(fb["slow"] / 1e6, 1) == 880.8 and round(fb["head"] / 1e6, 1) == 319.0
    assert abs(fb["total"] / 1e9 - 1.303) < 0.001
