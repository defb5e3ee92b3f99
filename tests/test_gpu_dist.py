"""GPU: the mesh-partitioned path through torch.distributed with real processes (tests/dist_worker.py under
torch.distributed.run) over gloo with both ranks on one GPU (runs everywhere); the same over RCCL with one GPU
per rank is tests/test_zy_rccl.py (needs two GPUs).  The stitched two-rank rollout, the loss and all 136 gradients must equal the
single-GPU run of the same mesh."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(backend, nproc=2):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MGN_DIST_BACKEND=backend, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("DIST_RESULT ")]
    assert line, res.stdout[-2000:] + res.stderr[-2000:]
    return json.loads(line[-1][len("DIST_RESULT "):])


def _check(r, world=2):
    assert r["finite"] and r["world"] == world
    assert r["rollout_rel"] < 1e-6, r
    assert r["loss_rel"] < 1e-5 and r["grad_rel_worst"] < 1e-5, r


def test_two_processes_gloo_transport_one_gpu():
    _check(_run("gloo"))


def test_one_process_rccl_backend():
    """The RCCL process group on the one GPU every box has: a single rank whose partition is the whole mesh (empty
    halos).  Runs the library initialisation, the all-reduces of statistics, loss and gradients and the (empty)
    all-to-all calls of the partitioned path on device buffers; the exchange between two GPUs is tests/test_zy_rccl.py."""
    _check(_run("nccl", nproc=1), world=1)
