#!/usr/bin/env python3
"""Diagnostic (GPU box): cycle shares between the STAMP() points of the fp16-split node kernel's
main loop (projections on).  Uses the -DMGN_PHASE_STAMPS copy of the library that ``python scripts/edge_bwd_stamps.py --build`` leaves in scripts/micro/.
    python scripts/phase_stamps_node.py [side]"""
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

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = gnn_amd.synthetic_frames(side - 1, side - 1)[0]
dev = "cuda:0"
n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
topo = gnn_amd.Topology(g["edge_index"], n, dev)
layer = gnn_amd.EdgeNodeMessagePassing(128).to(dev)
x, ef = torch.randn(n, 128, device=dev), torch.randn(e, 128, device=dev)
ps, pr, aggr = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
lib = _lib.lib()
lib.mgn_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
from gnn_amd import ops  # noqa: E402
nxt = gnn_amd.EdgeNodeMessagePassing(128).to(dev)
aggr.normal_()
torch.cuda.synchronize()
for it in range(3):
    lib.mgn_debug_read_stamps(buf, 1)               # clear
    ops.mp_node_project(x, aggr, topo.rowptr, layer.node_mlp.weights(), nxt.edge_mlp[0].weight.data, nxt.edge_mlp[0].bias.data,
                        x, ps, pr)
    torch.cuda.synchronize()
    lib.mgn_debug_read_stamps(buf, 0)
vals = list(buf)
names = {0: "end of pass -> top", 1: "stage: Xf, split x|aggr -> planes", 2: "barrier 1", 3: "prefetch next rows",
         4: "product 1 (K=256)", 5: "bias, ReLU, split -> Hp", 6: "barrier 2", 7: "product 2 + bias, ReLU",
         8: "LayerNorm (barrier inside)", 9: "residual, Os, split -> Np", 10: "barrier 4", 11: "x stores, projections -> Ps, Pr",
         12: "barrier 5", 13: "ps / pr stores"}
ntiles = (n + 31) // 32
for wsel, label in ((0, "wave 0"), (1, "wave 2")):
    tot = sum(vals[wsel * 16: wsel * 16 + 16])
    print(label, "cycles per tile: %.0f" % (tot / ntiles))
    for i in sorted(names):
        print("   %-84s %8.0f  %5.1f%%" % (names[i], vals[wsel * 16 + i] / ntiles, 100.0 * vals[wsel * 16 + i] / max(tot, 1)))
