#!/usr/bin/env python3
"""Bending-mesh rollout (33x17 nodes, 10 blocks) for a kernel trace: 100 steps, single trajectory, then 64 batched.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small -o small -- python3 scripts/prof_small.py [single|batched]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd  # noqa: E402
from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale  # noqa: E402
from gnn_amd.rollout import rollout, rollout_batched  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "single"
dev = "cuda:0"
torch.manual_seed(0)
m = EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=10, coarse_process_steps=0, node_out_dim=3,
                                  attention=False, with_mat_params=False, sample_ratio=0.0, time_dim=1, device=dev).to(dev).eval()
if mode == "single":
    frames = gnn_amd.synthetic_frames(32, 16, n_frames=100, seed=0, lx=0.08, ly=0.01)
    rollout(m, frames[:5])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rollout(m, frames)
    torch.cuda.synchronize(); print("single: %.3f ms/step" % ((time.perf_counter() - t0) * 10))
else:
    trajs = [gnn_amd.synthetic_frames(32, 16, n_frames=100, seed=s, lx=0.08, ly=0.01) for s in range(64)]
    rollout_batched(m, [t[:5] for t in trajs])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rollout_batched(m, trajs)
    torch.cuda.synchronize(); print("batched x64: %.3f ms/step" % ((time.perf_counter() - t0) * 10))
