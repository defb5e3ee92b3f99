#!/usr/bin/env python3
"""Edge build + receiver-major CSR, host routines against the device build (GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_amd.datasetclass import cells_to_edge_index, grid_mesh  # noqa: E402
from gnn_amd.graph import Topology  # noqa: E402

dev = "cuda:0"
for side in (317, 1000, 2000):
    _, cells = grid_mesh(side - 1, side - 1)
    n = side * side
    ct = torch.from_numpy(cells)
    t0 = time.perf_counter(); ei = cells_to_edge_index(ct); t1 = time.perf_counter()
    Topology(ei, n, dev); torch.cuda.synchronize(); t2 = time.perf_counter()
    cd = ct.to(dev)
    for _ in range(2):
        torch.cuda.synchronize(); t3 = time.perf_counter()
        eid = cells_to_edge_index(cd); torch.cuda.synchronize(); t4 = time.perf_counter()
        Topology(eid, n, dev); torch.cuda.synchronize(); t5 = time.perf_counter()
    assert torch.equal(eid.cpu(), ei)
    print(f"{n:8d} nodes {ei.shape[1]:9d} edges: host edges {1e3 * (t1 - t0):8.1f} ms + CSR and upload {1e3 * (t2 - t1):8.1f} ms;"
          f"  device edges {1e3 * (t4 - t3):6.2f} ms + CSR {1e3 * (t5 - t4):6.2f} ms")
