#!/usr/bin/env python3
"""Diagnostic (GPU box): cycle shares between the STAMP() points of the fused edge backward kernel.
``--build`` (build box) compiles a -DMGN_PHASE_STAMPS copy of the library to scripts/micro/libmgn_stamps.so; without it
(GPU box) the kernel runs from that copy.
    python scripts/edge_bwd_stamps.py --build;  python scripts/edge_bwd_stamps.py [side]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "scripts", "micro", "libmgn_stamps.so")
src = os.path.join(ROOT, "gnn_amd", "csrc")
if "--build" in sys.argv:          # on the build box: the stamped copy travels with the tree
    sys.path.insert(0, src)
    from build import SOURCES
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DMGN_PHASE_STAMPS",
                           "-o", out] + [os.path.join(src, f) for f in SOURCES], stderr=subprocess.DEVNULL)
    sys.exit(0)
from gnn_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("MGN_STAMPS_LIB", out)      # the product build: times only, all stamps zero
import gnn_amd  # noqa: E402
from gnn_amd import ops  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
side = int(args[0]) if args else 500
g = gnn_amd.synthetic_frames(side - 1, side - 1)[0]
dev = "cuda:0"
n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
topo = gnn_amd.Topology(g["edge_index"], n, dev)
layer = gnn_amd.EdgeNodeMessagePassing(128).to(dev)
em = layer.edge_mlp
f = lambda *s: torch.randn(*s, device=dev)
el, ge, gaggr, ps, pr = f(e, 128), 1e-3 * f(e, 128), 1e-3 * f(n, 128), f(n, 128), f(n, 128)
g1, dpr = torch.empty(e, 128, device=dev), torch.empty(n, 128, device=dev)
dw, db, dg, dbt = torch.empty(128, 128, device=dev), *(torch.empty(128, device=dev) for _ in range(3))
scratch = torch.empty(ops.train_scratch_bytes(), dtype=torch.uint8, device=dev)
lib = _lib.lib()
have_stamps = hasattr(lib, "mgn_debug_read_train_stamps")
if have_stamps:
    lib.mgn_debug_read_train_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
for it in range(6):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    ops.train_edge_bwd(el, ge, gaggr, ps, pr, topo.sender, topo.recv, topo.rowptr, em[0].weight.detach(), em[2].weight.detach(),
                       em[2].bias.detach(), em[4].weight.detach(), g1, dpr, dw, db, dg, dbt, scratch)
    t1.record(); torch.cuda.synchronize()
    if have_stamps:
        lib.mgn_debug_read_train_stamps(buf, 1)
print("kernel + reductions: %.3f ms" % t0.elapsed_time(t1))
if not have_stamps:
    sys.exit(0)
vals = list(buf)
names = {0: "barrier F -> top", 1: "requests", 2: "product 1", 3: "h1: ReLU, fp16 planes, bf16 planes", 4: "barrier B",
         5: "product 2, LN statistics, stage next e tile", 6: "barrier C", 7: "LN: xhat, upstream gradient, row sums", 8: "barrier C2",
         9: "g2 -> bf16 planes", 10: "barrier D", 11: "dW2 products", 12: "g1 product, mask, -> Ot", 13: "barrier E",
         14: "scan, row reads", 15: "g1 and dpr stores"}
ntiles = (e + 31) // 32
for wsel, label in ((0, "wave 0"), (1, "wave 2")):
    tot = sum(vals[wsel * 16: wsel * 16 + 16])
    print(label, "cycles per tile: %.0f" % (tot / ntiles))
    for i in range(16):
        print("   %-46s %8.0f  %5.1f%%" % (names[i], vals[wsel * 16 + i] / ntiles, 100.0 * vals[wsel * 16 + i] / max(tot, 1)))
