import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    """||a - b|| / ||b|| in float64."""
    import torch
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.fixture(params=["f16x2", "bf16x3", "f32"])
def matmul_mode(request):
    """Runs the test once per matrix path of the library (switched in this process, restored afterwards)."""
    from gnn_amd import ops
    prev = ops.set_matmul_mode(request.param)
    yield request.param
    ops.set_matmul_mode(prev)
