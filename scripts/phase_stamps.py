#!/usr/bin/env python3
"""Diagnostic (GPU box): cycle shares between the STAMP() points of the edge kernel's main loop, from the
-DMGN_PHASE_STAMPS copy of the library that ``python scripts/edge_bwd_stamps.py --build`` leaves in scripts/micro/.
    python scripts/phase_stamps.py [side]"""
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
for it in range(3):
    layer.step_csr(x, ef, topo, ps, pr, aggr)
    torch.cuda.synchronize()
    lib.mgn_debug_read_stamps(buf, 1)
vals = list(buf)
names = {0: "barrier E -> top", 1: "requests: gathers, next rows, ids", 2: "product 1", 3: "ReLU, split -> Hp", 4: "barrier B",
         5: "product 2, ReLU, LN statistics", 6: "barrier C", 7: "LN finish -> Os", 8: "barrier D", 9: "split + stage next tile",
         10: "scan, residual sums", 11: "barrier S", 12: "e_out and aggr stores"}
ntiles = (e + 31) // 32
for wsel, label in ((0, "wave 0"), (1, "wave 2")):
    tot = sum(vals[wsel * 16: wsel * 16 + 16])
    print(label, "cycles per tile (two workgroups share a CU: halve for the CU rate): %.0f" % (tot / ntiles))
    for i in range(13):
        print("   %-46s %8.0f  %5.1f%%" % (names[i], vals[wsel * 16 + i] / ntiles, 100.0 * vals[wsel * 16 + i] / max(tot, 1)))
