#!/usr/bin/env python3
"""Diagnostic (GPU box): cycle shares between the STAMP() points of the fp16-split node kernel."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "scripts", "micro", "libmgn_stamps.so")      # built by scripts/edge_bwd_stamps.py --build
from gnn_amd import _lib  # noqa: E402
_lib.LIB_PATH = out
import gnn_amd  # noqa: E402
from gnn_amd import ops  # noqa: E402
g = gnn_amd.synthetic_frames(999, 999)[0]
dev = "cuda:0"
n = g["mesh_pos"].shape[0]
topo = gnn_amd.Topology(g["edge_index"], n, dev)
layer = gnn_amd.EdgeNodeMessagePassing(128).to(dev)
x, aggr = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
xo, ps, pr = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
lib = _lib.lib()
lib.mgn_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
lib.mgn_debug_read_stamps(buf, 1)
nw = layer.node_mlp.weights()
for it in range(3):
    ops.mp_node_project(x, aggr, topo.rowptr, nw, layer.edge_mlp[0].weight.data, layer.edge_mlp[0].bias.data, xo, ps, pr)
    torch.cuda.synchronize()
    lib.mgn_debug_read_stamps(buf, 1)
vals = list(buf)
names = ["end of pass -> top", "stage: Xf, split x|aggr -> planes", "barrier 1", "prefetch next rows", "product 1 (K=256)", "bias, ReLU, split -> Hp",
         "barrier 2", "product 2 + bias, ReLU", "LayerNorm (barrier inside)", "residual, Os, split -> Np", "barrier 4", "projections -> Ps, Pr",
         "barrier 5", "stores"]
ntiles = (n + 31) // 32
for wsel, label in ((0, "wave 0"), (1, "wave 2")):
    tot = sum(vals[wsel * 16: wsel * 16 + 16])
    print(label, "cycles per tile: %.0f" % (tot / ntiles))
    for i, nm in enumerate(names):
        print("   %-40s %8.0f  %5.1f%%" % (nm, vals[wsel * 16 + i] / ntiles, 100.0 * vals[wsel * 16 + i] / max(tot, 1)))
