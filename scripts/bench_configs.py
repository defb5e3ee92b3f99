#!/usr/bin/env python3
"""Secondary configurations of the baseline (GPU box): bending-mesh rollout (33x17 nodes, 10 MP
steps, real bending normaliser statistics if given) and the 100k-node mesh with 15 MP steps,
optionally B block-diagonal copies (batched trajectories on one topology).
    python scripts/bench_configs.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd  # noqa: E402
from gnn_amd.graph import MeshGraph  # noqa: E402
from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale  # noqa: E402
from gnn_amd.rollout import rollout  # noqa: E402

dev = "cuda:0"


def model_for(steps):
    torch.manual_seed(0)
    return EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=steps, coarse_process_steps=0, node_out_dim=3,
                                         attention=False, with_mat_params=False, sample_ratio=0.0, time_dim=1,
                                         device=dev).to(dev).eval()


def batched(frames, copies):
    """Block-diagonal union of `copies` independent states on one mesh topology."""
    return gnn_amd.batch_graphs([frames[0]] * copies)


# --- bending rollout, 200 frames
frames = gnn_amd.synthetic_frames(32, 16, n_frames=200, seed=0, lx=0.08, ly=0.01)
m = model_for(10)
rollout(m, frames[:20])
torch.cuda.synchronize(); t0 = time.perf_counter()
rollout(m, frames)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"bending 33x17 rollout: {200 / dt:.1f} steps/s ({dt / 200 * 1e3:.3f} ms/step), {3168 * 10 * 200 / dt / 1e6:.1f} M edge-updates/s")

# --- 64 bending trajectories advanced together
from gnn_amd.rollout import rollout_batched  # noqa: E402
trajs = [gnn_amd.synthetic_frames(32, 16, n_frames=200, seed=s, lx=0.08, ly=0.01) for s in range(64)]
rollout_batched(m, [t[:10] for t in trajs])
torch.cuda.synchronize(); t0 = time.perf_counter()
rollout_batched(m, trajs)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"bending 33x17 x64 trajectories: {64 * 200 / dt:.0f} trajectory-steps/s ({dt / 200 * 1e3:.3f} ms per batched step, "
      f"host-side frame merging included)")

# --- 100k mesh, 15 MP steps, 1 and 8 trajectories
for copies in (1, 8):
    g = batched(gnn_amd.synthetic_frames(316, 316, seed=0), copies).to(dev)
    m = model_for(15)
    e = g["edge_index"].shape[1]
    for _ in range(2):
        m.predict(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        m.predict(g)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"100k mesh x{copies}: predict {dt * 1e3:.2f} ms, {copies / dt:.1f} trajectory-steps/s, {e * 15 / dt / 1e6:.0f} M edge-updates/s")

# --- bending-size training step (loss + backward + Adam), the size of the reference's own training log
from gnn_amd.optim import FusedAdam  # noqa: E402
fr = gnn_amd.synthetic_frames(32, 16, n_frames=1, seed=0, lx=0.08, ly=0.01)[0]
nb = fr["mesh_pos"].shape[0]
fr["target"] = torch.cat([fr["world_pos"], fr["phi"]], -1).unsqueeze(0) + 1e-4 * torch.randn(1, nb, 3)
fr = fr.to(dev)
m = model_for(10).train()
opt = FusedAdam(m.parameters(), lr=1e-4)


def train_step():
    opt.zero_grad()
    m.loss(fr)[0].backward()
    opt.step()


for _ in range(5):
    train_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    train_step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"bending 33x17 training step (loss + backward + Adam): {1 / dt:.1f} steps/s ({dt * 1e3:.2f} ms)")

# --- the same step replayed from a captured HIP graph
from gnn_amd.training import CapturedTrainStep  # noqa: E402
m = model_for(10).train()
opt = FusedAdam(m.parameters(), lr=1e-4, capturable=True)
stepper = CapturedTrainStep(m, opt)
for _ in range(5):
    stepper(fr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    stepper(fr)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"bending 33x17 training step, captured graph: {1 / dt:.1f} steps/s ({dt * 1e3:.2f} ms)")
