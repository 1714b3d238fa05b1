"""The driver's smoke entry point, run the way the driver runs it (its own process: the engine needs the device's engine
slot, and the process state is the one the frame engine's fits-a-CU gate once answered differently in: the runtime's
occupancy query returned 0 for a kernel that runs - engine.hip: eng_fits_cu)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smoke_entry_point_runs_in_its_own_process():
    env = dict(os.environ)
    env.pop("FT_NO_ENGINE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "smoke"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "smoke ok" in r.stdout and "frame-engine frames" in r.stdout, r.stdout[-2000:]
