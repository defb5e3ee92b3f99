#!/usr/bin/env python3
"""Same-box A/B of the processor kernels: times mp_edge and mp_node(+projections) at 1M nodes with the library
named by MGN_LIB_PATH (default: the in-tree one).  python scripts/ab_node.py [side] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd  # noqa: E402
from gnn_amd import ops  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda:0"
g = gnn_amd.synthetic_frames(side - 1, side - 1)[0]
n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
topo = gnn_amd.Topology(g["edge_index"], n, dev)
torch.manual_seed(0)
layer, nxt = gnn_amd.EdgeNodeMessagePassing(128).to(dev), gnn_amd.EdgeNodeMessagePassing(128).to(dev)
x, ef = torch.randn(n, 128, device=dev), torch.randn(e, 128, device=dev)
ps, pr, aggr = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
pk = ops.PackedWeights().refresh([layer, nxt])
x2 = torch.empty_like(x)


def node():
    ops.mp_node_packed(x, aggr, topo.rowptr, layer.node_mlp.weights(), pk.node[0], x2, nxt.edge_mlp[0].weight.data,
                       nxt.edge_mlp[0].bias.data, ps, pr)


def edge():
    ops.mp_edge(ef, ps, pr, topo.rowptr, topo.sender, topo.recv, layer.edge_mlp.weights(), ef, aggr, packed=pk.edge[0])


ef2 = torch.empty_like(ef)


def edge_out_of_place():
    ops.mp_edge(ef, ps, pr, topo.rowptr, topo.sender, topo.recv, layer.edge_mlp.weights(), ef2, aggr, packed=pk.edge[0])


def edge_no_store():
    ops.mp_edge(ef, ps, pr, topo.rowptr, topo.sender, topo.recv, layer.edge_mlp.weights(), None, aggr, packed=pk.edge[0])


for name, fn in (("node", node), ("edge", edge), ("node", node), ("edge, e_out to a second buffer", edge_out_of_place),
                 ("edge", edge), ("edge, e_out not stored", edge_no_store)):
    for _ in range(3):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    t.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('MGN_LIB_PATH', 'in-tree'):40s} {name}: {s.elapsed_time(t) / reps:.4f} ms", flush=True)
print("checksum", float(x2.double().sum()), float(ps.double().sum()))
