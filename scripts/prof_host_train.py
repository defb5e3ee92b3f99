#!/usr/bin/env python3
"""Host-side profile (cProfile) of the bending-mesh training step: where the interpreter spends a step."""
import cProfile
import os
import pstats
import sys

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
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
