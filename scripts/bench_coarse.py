"""Time of the coarse-graph build (farthest-point sampling + kNN graph + receiver CSR) on the device against the
host functions, per mesh size:  python scripts/bench_coarse.py [ratio]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from gnn_amd import ops                                   # noqa: E402
from gnn_amd.coarse import CoarseGraph, farthest_point_sampling, knn_graph   # noqa: E402

ratio = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
dev = "cuda:0"
for side, host in ((24, True), (100, True), (317, True), (1000, False)):
    xs, ys = torch.meshgrid(torch.linspace(0, 1, side), torch.linspace(0, 1, side), indexing="xy")
    pts = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).contiguous()
    n, m = pts.shape[0], max(1, int(pts.shape[0] * ratio))
    if side == 1000:
        m = 20000                                          # sampling rounds are sequential: a bounded sample at 1M points
    d = pts.to(dev)
    ops.fps(d, [n], [min(m, 8)], [0])                      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = ops.fps(d, [n], [m], [0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ei = ops.knn_graph(d, first, [m], 3)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    line = f"{n:8d} nodes, {m:7d} samples: device sampling {1e3 * (t1 - t0):9.2f} ms ({1e6 * (t1 - t0) / m:6.2f} us per sample), kNN {1e3 * (t2 - t1):8.2f} ms"
    if host:
        t0 = time.perf_counter()
        hf = farthest_point_sampling(pts, m, 0)
        t1 = time.perf_counter()
        he = knn_graph(pts[hf], 3)
        t2 = time.perf_counter()
        same = torch.equal(hf, first.cpu()) and torch.equal(he, ei.cpu())
        line += f" | host sampling {1e3 * (t1 - t0):9.1f} ms, kNN {1e3 * (t2 - t1):8.1f} ms, identical: {same}"
    print(line, flush=True)
for side in (24, 317):
    xs, ys = torch.meshgrid(torch.linspace(0, 1, side), torch.linspace(0, 1, side), indexing="xy")
    d = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).contiguous().to(dev)
    CoarseGraph(d, ratio, 0, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    CoarseGraph(d, ratio, 1, dev)
    torch.cuda.synchronize()
    print(f"{side * side:8d} nodes: whole coarse graph (sampling, kNN, CSR, end-point ids) {1e3 * (time.perf_counter() - t0):.2f} ms")
