"""GPU: the device edge build (cells -> sorted unique directed edges) and the device receiver-major
CSR against the reference's golden edge lists, the oracle and the host routines of the same
library -- integer work, so everything is compared for exact equality."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_device_edge_build_matches_golden_and_oracle():
    from gnn_amd.datasetclass import cells_to_edge_index
    d = np.load(f"{GOLDEN}/g1_edges.npz")
    for name in ("g2x2", "g4x3", "g32x16", "irregular", "two_tris"):
        ei = cells_to_edge_index(torch.from_numpy(d[f"{name}_cells"]).to(DEV))
        assert ei.is_cuda and ei.dtype == torch.int64
        assert np.array_equal(ei.cpu().numpy(), d[f"{name}_edge_index"]), name
    rng = np.random.default_rng(0)
    for _ in range(5):                                   # random soups of triangles (duplicates, degenerate ones)
        cells = rng.integers(0, 1000, size=(rng.integers(1, 400), 3))
        got = cells_to_edge_index(torch.from_numpy(cells).to(DEV)).cpu().numpy()
        assert np.array_equal(got, O.cells_to_edge_index(cells))
    assert cells_to_edge_index(torch.zeros(0, 3, dtype=torch.long, device=DEV)).shape == (2, 0)
    with pytest.raises(RuntimeError, match="negative"):
        cells_to_edge_index(torch.tensor([[0, 1, -2]], device=DEV))


def test_device_edge_build_full_size_grid():
    """The benchmark mesh (1000 x 1000 nodes): device and host builds agree edge for edge; ids near
    the top of the range exercise the high key bits."""
    from gnn_amd.datasetclass import cells_to_edge_index, grid_mesh
    _, cells = grid_mesh(999, 999)
    host = cells_to_edge_index(torch.from_numpy(cells))
    dev = cells_to_edge_index(torch.from_numpy(cells).to(DEV))
    assert host.shape == (2, 5992002) and torch.equal(dev.cpu(), host)
    big = torch.tensor([[0, 2 ** 31 - 2, 2 ** 31 - 1], [5, 2 ** 31 - 1, 7]])
    assert torch.equal(cells_to_edge_index(big.to(DEV)).cpu(), cells_to_edge_index(big))


def test_device_receiver_csr_matches_host():
    from gnn_amd import ops
    rng = np.random.default_rng(1)
    for n, e in ((1, 0), (5, 3), (50, 400), (1000, 9000), (70000, 300000)):
        ei = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)])
        if n > 10:
            ei[1][ei[1] == 3] = 4                          # node 3 receives nothing
        exp = ops.receiver_csr_np(ei, n)
        got = ops.receiver_csr_dev(torch.from_numpy(ei).to(DEV), n)
        for a, b in zip(got, exp):
            assert a.is_cuda and np.array_equal(a.cpu().numpy(), b) and a.cpu().numpy().dtype == b.dtype
    with pytest.raises(RuntimeError, match="outside"):
        ops.receiver_csr_dev(torch.tensor([[0, 5], [1, 1]], device=DEV), 3)
    with pytest.raises(RuntimeError, match="outside"):
        ops.receiver_csr_dev(torch.tensor([[0, 1], [1, -1]], device=DEV), 3)


def test_topology_is_the_same_from_either_side():
    from gnn_amd.graph import Topology
    g = O.synthetic_graph(40, 30, seed=2)
    n = g["mesh_pos"].shape[0]
    a = Topology(g["edge_index"], n, DEV)                  # host build, uploaded
    b = Topology(g["edge_index"].to(DEV), n, DEV)          # device build
    for k in ("rowptr", "sender", "recv", "perm"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a.max_in_degree == b.max_in_degree and a.num_edges == b.num_edges
