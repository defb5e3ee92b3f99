#!/usr/bin/env python3
"""Bending-mesh training step (33x17 nodes, 10 blocks: loss + backward + fused Adam) for a kernel trace.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small -o train -- python3 scripts/prof_small_train.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd  # noqa: E402
from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale  # noqa: E402
from gnn_amd.optim import FusedAdam  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=10, coarse_process_steps=0, node_out_dim=3,
                                  attention=False, with_mat_params=False, sample_ratio=0.0, time_dim=1, device=dev).to(dev).train()
fr = gnn_amd.synthetic_frames(32, 16, n_frames=1, seed=0, lx=0.08, ly=0.01)[0]
n = fr["mesh_pos"].shape[0]
fr["target"] = torch.cat([fr["world_pos"], fr["phi"]], -1).unsqueeze(0) + 1e-4 * torch.randn(1, n, 3)
fr = fr.to(dev)
opt = FusedAdam(m.parameters(), lr=1e-4)


def step():
    opt.zero_grad()
    m.loss(fr)[0].backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(60):
    step()
torch.cuda.synchronize(); print("training step: %.3f ms" % ((time.perf_counter() - t0) / 60 * 1e3))
