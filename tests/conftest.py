import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def layer_taps():
    """The wave-per-sample layer kernels keep a layer's intermediates in registers (production saves only the layer inputs);
    V4L_LAYER_TAPS=1 makes the SAME kernels also write every intermediate row-major into the workspace slots the tap tests
    read through v4l_net_ws_offset (read per call by the library)."""
    os.environ["V4L_LAYER_TAPS"] = "1"
    yield
    os.environ.pop("V4L_LAYER_TAPS", None)


def pytest_sessionfinish(session, exitstatus):
    """Keep the parity errors the GPU tests measured (util.record) as a file: gpurun_out/parity.json."""
    import json
    import util
    if not util.PARITY:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity.json"), "w") as f:
            json.dump({"exitstatus": int(exitstatus), "errors": dict(sorted(util.PARITY.items()))}, f, indent=1)
    except OSError:
        pass
