#!/usr/bin/env python3
"""Diagnostic (GPU box): cycle shares between the STAMP() points of the split-bf16 edge kernel's
main loop.  Builds a -DMGN_PHASE_STAMPS copy of the library into gpurun_out/ and runs one layer.
    MGN_MATMUL=bf16x3 python scripts/phase_stamps.py [side]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libmgn_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
src = os.path.join(ROOT, "gnn_amd", "csrc")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DMGN_PHASE_STAMPS",
                       "-o", out, os.path.join(src, "mgn_kernels.hip"), os.path.join(src, "mgn_train.hip"), os.path.join(src, "mgn_host.cpp")])
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
names0 = {0: "wait b2 -> top", 6: "issue loads", 1: "product 1a", 2: "wait bm", 3: "product 1b", 4: "wait b1", 5: "ReLU/split -> Hp, stage next"}
names1 = {0: "wait b2 -> top", 1: "LN finish -> Es", 2: "wait bm", 3: "store + scan", 4: "wait b1", 6: "scan finish", 5: "product 2 + stats"}
ntiles = (e + 31) // 32
for wsel, label, names in ((0, "role 0, wave 0", names0), (1, "role 1, wave 4", names1)):
    tot = sum(vals[wsel * 16: wsel * 16 + 16])
    print(label, "cycles per tile: %.0f" % (tot / ntiles))
    for i in sorted(names, key=lambda q: [0, 6, 1, 2, 3, 4, 5].index(q) if wsel == 0 else [0, 1, 2, 3, 4, 6, 5].index(q)):
        print("   %-30s %8.0f  %5.1f%%" % (names[i], vals[wsel * 16 + i] / ntiles, 100.0 * vals[wsel * 16 + i] / max(tot, 1)))
