#!/usr/bin/env python3
"""Diagnostic (GPU box): edge-kernel time on meshes of equal size and different row length.  Short
rows put the three mesh rows a workgroup gathers sender projections from into L2; the difference to
the square mesh is what a locality-aware tile order could gain there."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_amd  # noqa: E402
from gnn_amd import ops  # noqa: E402

dev = "cuda:0"
layer = gnn_amd.EdgeNodeMessagePassing(128).to(dev)
ew = layer.edge_mlp.weights()
for nx, ny in ((1000, 1000), (250, 4000), (100, 10000), (32, 31250)):
    g = gnn_amd.synthetic_frames(nx - 1, ny - 1)[0]
    n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
    topo = gnn_amd.Topology(g["edge_index"], n, dev)
    ps, pr, aggr = (torch.randn(n, 128, device=dev) for _ in range(3))
    ein, eout = torch.randn(e, 128, device=dev), torch.empty(e, 128, device=dev)
    ts = []
    for it in range(8):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        ops.mp_edge(ein, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, eout, aggr)
        t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1))
    print(f"{nx:5d} x {ny:6d}  N={n} E={e}  edge kernel {min(ts[2:]):.3f} ms (median {sorted(ts[2:])[3]:.3f})  {e / min(ts[2:]) * 1e-6:.2f} G edges/s")
