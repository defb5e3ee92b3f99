import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gnn_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = sys.argv[1]
import gnn_amd
g = gnn_amd.synthetic_frames(999, 999)[0]; dev = "cuda:0"
n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
topo = gnn_amd.Topology(g["edge_index"], n, dev); layer = gnn_amd.EdgeNodeMessagePassing(128).to(dev)
x, ef = torch.randn(n, 128, device=dev), torch.randn(e, 128, device=dev)
ps, pr, aggr = torch.randn_like(x), torch.randn_like(x), torch.empty_like(x)
from gnn_amd import ops
ew = layer.edge_mlp.weights()
for _ in range(2): ops.mp_edge(ef, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, ef, aggr)
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): ops.mp_edge(ef, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, ef, aggr)
t.record(); torch.cuda.synchronize()
print(os.path.basename(_lib.LIB_PATH), "edge kernel ms:", s.elapsed_time(t) / 5)
nw = layer.node_mlp.weights()
w0, b0 = layer.edge_mlp[0].weight.data, layer.edge_mlp[0].bias.data
xo = torch.empty_like(x)
for _ in range(2): ops.mp_node_project(x, aggr, topo.rowptr, nw, w0, b0, xo, ps, pr)
s.record()
for _ in range(5): ops.mp_node_project(x, aggr, topo.rowptr, nw, w0, b0, xo, ps, pr)
t.record(); torch.cuda.synchronize()
print(os.path.basename(_lib.LIB_PATH), "node+project kernel ms:", s.elapsed_time(t) / 5)
