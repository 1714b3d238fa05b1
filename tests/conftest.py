import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """A GPU test that FAILS leaves its engines open (the traceback keeps them alive), and an open batch-1 engine holds its
    device's frame-engine seat: every later test of the run would then find the seat taken and fail for that reason alone.
    Engines created during the call phase of a failed test are closed here (fixtures are set up in another phase and
    stay)."""
    if "gpu" not in item.keywords:
        yield
        return
    from fish_tts_amd.ar_engine import ARHipEngine
    created = []
    init = ARHipEngine.__init__

    def recording_init(self, *a, **kw):
        created.append(self)
        init(self, *a, **kw)
    ARHipEngine.__init__ = recording_init
    try:
        outcome = yield
    finally:
        ARHipEngine.__init__ = init
    if outcome.excinfo is not None:
        for eng in created:
            try:
                eng.close()
            except Exception:  # noqa: BLE001
                pass
