"""GPU: the two-process test of tests/test_gpu_dist.py over RCCL, one GPU per rank (skipped on a one-GPU box).
In a file of its own that is collected late: it is the one test that cannot run on the one-GPU development box."""
import pytest
import torch

from test_gpu_dist import _check, _run

pytestmark = pytest.mark.gpu


def test_two_processes_rccl_one_gpu_each():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU over RCCL)")
    _check(_run("nccl"))
