#!/usr/bin/env python3
"""Diagnostic (GPU box): host-side profile of the eager launch sequence of one small-mesh predict."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_amd  # noqa: E402
from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=10, coarse_process_steps=0, node_out_dim=3,
                                  attention=False, with_mat_params=False, sample_ratio=0.0, time_dim=1, device=dev).to(dev).eval()
g = gnn_amd.synthetic_frames(32, 16, n_frames=1, seed=0)[0].to(dev)
for _ in range(20):
    m.predict(g)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    m.predict(g)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

# ---- the same for a training step
from gnn_amd.optim import FusedAdam  # noqa: E402
g["target"] = torch.cat([g["world_pos"], g["phi"]], -1).unsqueeze(0) + 1e-4 * torch.randn(1, g["mesh_pos"].shape[0], 3, device=dev)
m.train()
opt = FusedAdam(m.parameters(), lr=1e-4)


def train_step():
    opt.zero_grad()
    m.loss(g)[0].backward()
    opt.step()


for _ in range(10):
    train_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    train_step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
