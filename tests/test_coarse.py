"""Host side of the coarse multiscale branch (core/meshgraphnet.py:237-283): farthest-point sampling and the
k-nearest-neighbour graph of gnn_amd/coarse.py against fixture G7 and against the oracle's restatement."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import GOLDEN
from gnn_amd.coarse import farthest_point_sampling, knn_graph


@pytest.mark.parametrize("tag,ratio", [("ms", 0.5), ("both", 0.4)])
def test_sample_and_knn_graph_golden(tag, ratio):
    d = np.load(f"{GOLDEN}/g7_attention_multiscale.npz")
    g = O.synthetic_graph(int(d["nx"]), int(d["ny"]), seed=int(d["graph_seed"]))
    pts = g["mesh_pos"].float()
    first = farthest_point_sampling(pts, int(pts.shape[0] * ratio), int(d[f"{tag}_start1"]))
    assert np.array_equal(first.numpy(), d[f"{tag}_first1"])
    assert np.array_equal(knn_graph(pts[first], 3).numpy(), d[f"{tag}_coarse_edge_index1"])


@pytest.mark.parametrize("n,dim,k,seed", [(1, 2, 3, 0), (2, 2, 3, 0), (4, 2, 3, 1), (300, 2, 3, 2), (257, 3, 4, 3)])
def test_knn_graph_equals_oracle_on_point_clouds(n, dim, k, seed):
    """Random clouds (no ties) and degenerate sizes (fewer points than k + 1)."""
    pts = torch.rand(n, dim, generator=torch.Generator().manual_seed(seed))
    assert torch.equal(knn_graph(pts, k), O.knn_graph(pts, k))


def test_knn_graph_ties_on_a_regular_grid():
    """Every interior point of a unit grid has four neighbours at distance 1: k = 3 keeps the three lowest
    indices; the candidate window is widened until no tie is cut."""
    xs, ys = torch.meshgrid(torch.arange(9.0), torch.arange(7.0), indexing="ij")
    pts = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    ei = knn_graph(pts, 3)
    assert torch.equal(ei, O.knn_graph(pts, 3))
    c = 4 * 7 + 3                                       # interior point (4, 3)
    assert ei[0, ei[1] == c].tolist() == [c - 7, c - 1, c + 1]
    # all points coincide: every distance ties, the k lowest other indices win
    ei0 = knn_graph(torch.zeros(40, 2), 3)
    assert torch.equal(ei0, O.knn_graph(torch.zeros(40, 2), 3))
    assert ei0[0, ei0[1] == 20].tolist() == [0, 1, 2] and ei0[0, ei0[1] == 0].tolist() == [1, 2, 3]
