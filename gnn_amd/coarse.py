"""Coarse graph of the multiscale option (core/meshgraphnet.py:237-283): farthest-point sampling of the mesh
nodes and a k-nearest-neighbour graph over the sample.  Host-side topology preparation, like the edge build; the
arithmetic on the coarse graph (edge features, encoder, processor blocks) runs in the HIP kernels.

The reference rebuilds this graph inside every ``forward`` with a random first sample (:248); here a graph is
cached per (mesh, first sample), so a fixed first sample (``model.coarse_start``) costs the build once.
``knn_graph`` comes from torch-cluster in the reference (absent in this image, no vector pins it): restated as
"k nearest other points, nearest first, ties by lower index, edges (neighbour -> point) grouped by point"."""
from __future__ import annotations

import numpy as np
import torch

from .graph import Topology


def farthest_point_sampling(points: torch.Tensor, num_samples: int, start: int) -> torch.Tensor:
    """Indices of ``num_samples`` points, each the farthest from those chosen before it (:243-254); ``points`` on
    the CPU, float32 -- the same torch expressions as the reference, hence the same choices at equal distances."""
    n = points.shape[0]
    idx = torch.zeros(num_samples, dtype=torch.long)
    idx[0] = int(start)
    dist = torch.full((n,), float("inf"))
    for i in range(1, num_samples):
        last = points[idx[i - 1]].unsqueeze(0)
        dist = torch.minimum(dist, torch.norm(points - last, dim=-1))
        idx[i] = torch.argmax(dist)
    return idx


def knn_graph(points: torch.Tensor, k: int) -> torch.Tensor:
    """[2, n * min(k, n-1)] int64: row 0 = neighbour (sender), row 1 = point (receiver)."""
    from scipy.spatial import cKDTree
    p = points.detach().cpu().double().numpy()
    n = p.shape[0]
    kk = min(k, n - 1)
    if kk <= 0:
        return torch.zeros(2, 0, dtype=torch.long)
    cand = min(n, kk + 17)                                    # room for ties at the k-th distance
    while True:
        _, nb = cKDTree(p).query(p, k=cand)
        nb = nb.reshape(n, cand)
        diff = p[nb] - p[:, None, :]
        d2 = diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1] if p.shape[1] == 2 else (diff * diff).sum(-1)
        d2 = np.where(nb == np.arange(n)[:, None], np.inf, d2)
        order = np.lexsort((nb, d2), axis=1)                  # by squared distance, then by index
        nb_s, d2_s = np.take_along_axis(nb, order, 1), np.take_along_axis(d2, order, 1)
        # a tie that reaches the end of the candidate list may hide a lower index outside it: widen and redo
        if cand == n or not np.any(d2_s[:, kk - 1] == d2_s[:, cand - 2]):
            break
        cand = min(n, 2 * cand)
    centre = np.repeat(np.arange(n), kk)
    return torch.from_numpy(np.stack([nb_s[:, :kk].reshape(-1), centre]).astype(np.int64))


class CoarseGraph:
    """Sample, its kNN graph in receiver-major order on the device, and the original node ids of the coarse
    edges' end points (for the edge features, which are built from the fine mesh's fields, :276-281)."""

    def __init__(self, mesh_pos: torch.Tensor, sample_ratio: float, start: int, device):
        pts = mesh_pos.detach().to("cpu", torch.float32)
        k = 4 if pts.shape[-1] == 3 else 3
        n_coarse = max(1, int(pts.shape[0] * sample_ratio))
        self.start = int(start)
        self.first_cpu = farthest_point_sampling(pts, n_coarse, start)
        self.edge_index = knn_graph(pts[self.first_cpu], k)                # coarse numbering
        self.num_nodes = n_coarse
        self.first = self.first_cpu.to(device)                             # int64, for row gather / scatter
        self.topology = Topology(self.edge_index, n_coarse, device)
        first32 = self.first.to(torch.int32)
        self.sender_orig = first32[self.topology.sender.long()].contiguous()
        self.recv_orig = first32[self.topology.recv.long()].contiguous()
