#!/usr/bin/env python3
"""Diagnostic (GPU box): times the edge-kernel variants of scripts/micro/edge_variants.hip on the 1M-node
mesh, interleaved rounds in one process, and compares the outputs of the non-ablated ones with the product
kernel's.  Build the sandbox library on the CPU box first (command in the .hip file's header).
    python scripts/exp_edge_variants.py [variant ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd  # noqa: E402
from gnn_amd import ops  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "scripts", "micro", "libedge_variants.so"))
P = C.c_void_p
lib.edge_variant.argtypes = [C.c_int] + [P] * 13 + [C.c_int, C.c_int, P]
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 8, 12, 32, 64, 14, 78, 110]
exact = {int(v) for v in os.environ.get("EXACT", "0").split(",")}     # variants whose output must match the product kernel
side = int(os.environ.get("SIDE", "1000"))

dev = "cuda:0"
torch.manual_seed(0)
g = gnn_amd.synthetic_frames(side - 1, side - 1)[0]
n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
topo = gnn_amd.Topology(g["edge_index"], n, dev)
layer = gnn_amd.EdgeNodeMessagePassing(128).to(dev)
ew = layer.edge_mlp.weights()
w0, b0, w1, b1, lnw, lnb = ew.tensors
ps, pr = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
ein = torch.randn(e, 128, device=dev)
eout_ref, aggr_ref = torch.empty(e, 128, device=dev), torch.zeros(n, 128, device=dev)
ops.mp_edge(ein, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, eout_ref, aggr_ref)
eout, aggr = torch.empty(e, 128, device=dev), torch.zeros(n, 128, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def run(v):
    rc = lib.edge_variant(v, ein.data_ptr(), ps.data_ptr(), pr.data_ptr(), topo.rowptr.data_ptr(), topo.sender.data_ptr(),
                          topo.recv.data_ptr(), w0.data_ptr(), w1.data_ptr(), b1.data_ptr(), lnw.data_ptr(), lnb.data_ptr(),
                          eout.data_ptr(), aggr.data_ptr(), e, n, stream)
    assert rc == 0, (v, rc)


times = {v: [] for v in variants}
times["product"] = []
for rnd in range(7):
    for v in ["product"] + variants:
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        if v == "product":
            ops.mp_edge(ein, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, eout, aggr)
        else:
            run(v)
        t1.record()
        torch.cuda.synchronize()
        times[v].append(t0.elapsed_time(t1))
        if rnd == 0 and v in exact:
            de, da = float((eout - eout_ref).abs().max()), float((aggr - aggr_ref).abs().max())
            print(f"variant {v}: max |e_out - product| = {de:.3e}, max |aggr - product| = {da:.3e}", flush=True)
for v, ts in times.items():
    ts = sorted(ts[1:])
    print(f"variant {str(v):>8s}: min {ts[0]:.3f} ms  median {ts[len(ts) // 2]:.3f} ms", flush=True)
