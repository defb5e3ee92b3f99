"""GPU: the C++ host example that drives a partitioned message-passing stack through the C ABI alone (no Python, no
torch in the process): built in-tree by ``gnn_amd/csrc/build.py``, run here as a child process."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "partitioned_step")


@pytest.mark.parametrize("args", [[], ["33", "17", "2"], ["120", "90", "8"], ["9", "7", "1"]], ids=["60x45x3", "33x17x2", "120x90x8", "9x7x1"])
def test_partitioned_step_example_matches_the_whole_mesh(args):
    if not os.path.exists(EXE):
        pytest.fail("examples/partitioned_step is missing: build it with `python gnn_amd/csrc/build.py`")
    res = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=120)
    print(res.stdout[-800:], res.stderr[-800:])
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), (res.returncode, res.stdout[-400:], res.stderr[-400:])
