"""Coarse graph of the multiscale option (core/meshgraphnet.py:237-283): farthest-point sampling of the mesh
nodes and a k-nearest-neighbour graph over the sample.  For a mesh that lives on the GPU both run in the library
(``mgn_fps``, ``mgn_knn_graph``) and the graph never leaves the device; the two host functions of this module are the
same definitions for CPU tensors (like the host edge build) and give the same indices.  The arithmetic on the coarse
graph (edge features, encoder, processor blocks) runs in the HIP kernels either way.

The reference rebuilds this graph inside every ``forward`` with a random first sample (:248); here a graph is
cached per (mesh, first sample), so a fixed first sample (``model.coarse_start``) costs the build once.
``knn_graph`` comes from torch-cluster in the reference (absent in this image, no vector pins it): restated as
"k nearest other points, nearest first, ties by lower index, edges (neighbour -> point) grouped by point".

A block-diagonal batch of meshes (``batch_sizes``) is sampled and connected block by block, so every block gets the
coarse graph it would get alone."""
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


def sample_and_connect(mesh_pos: torch.Tensor, sample_ratio: float, starts, batch_sizes=None):
    """(first [M] int64, edge_index [2, Ec] int64 in coarse numbering, coarse block sizes): the sampled nodes and their
    kNN graph, block by block.  ``starts``: the first sample of every block as a row of that block (:248).  On the
    device of ``mesh_pos``: library kernels for a GPU tensor, the host functions above for a CPU tensor."""
    from . import ops
    pts = mesh_pos.detach().to(torch.float32).contiguous()
    sizes = [int(pts.shape[0])] if batch_sizes is None else [int(b) for b in batch_sizes]
    if sum(sizes) != pts.shape[0]:
        raise RuntimeError("coarse graph: batch sizes do not add up to the number of nodes")
    k = 4 if pts.shape[-1] == 3 else 3                                     # :259
    counts = [max(1, int(n * sample_ratio)) for n in sizes]                # :261
    starts = [int(s) for s in starts]
    if len(starts) != len(sizes) or any(not 0 <= s < n for s, n in zip(starts, sizes)):
        raise RuntimeError("coarse graph: one first sample per block, inside the block")
    if pts.is_cuda:
        first = ops.fps(pts, sizes, counts, starts)
        return first, ops.knn_graph(pts, first, counts, k), counts
    firsts, edges, n0, c0 = [], [], 0, 0
    for n, m, st in zip(sizes, counts, starts):
        f = farthest_point_sampling(pts[n0: n0 + n], m, st)
        edges.append(knn_graph(pts[n0: n0 + n][f], k) + c0)
        firsts.append(f + n0)
        n0, c0 = n0 + n, c0 + m
    return torch.cat(firsts), torch.cat(edges, dim=1), counts


class CoarseGraph:
    """Sample, its kNN graph in receiver-major order on the device, and the original node ids of the coarse
    edges' end points (for the edge features, which are built from the fine mesh's fields, :276-281).
    ``start``: the first sample (row of the mesh), or one per block of ``batch_sizes``."""

    def __init__(self, mesh_pos: torch.Tensor, sample_ratio: float, start, device, batch_sizes=None):
        starts = [int(s) for s in start] if isinstance(start, (list, tuple)) else [int(start)] * (len(batch_sizes) if batch_sizes else 1)
        self.start = starts[0] if len(starts) == 1 else tuple(starts)
        first, self.edge_index, counts = sample_and_connect(mesh_pos, sample_ratio, starts, batch_sizes)
        n_coarse = sum(counts)
        self.num_nodes, self.block_sizes = n_coarse, counts
        self.first = first.to(device)                                      # int64, for row gather / scatter
        # both builders emit the edges grouped by point, points ascending: that is receiver-major order already, with
        # min(k, block size - 1) in-edges per point
        k = 4 if mesh_pos.shape[-1] == 3 else 3
        degree = np.concatenate([np.full(m, min(k, m - 1), dtype=np.int64) for m in counts]) if counts else np.zeros(0, np.int64)
        rowptr = np.zeros(n_coarse + 1, dtype=np.int64)
        np.cumsum(degree, out=rowptr[1:])
        if rowptr[-1] != self.edge_index.shape[1]:
            raise RuntimeError("coarse graph: the nearest-neighbour graph has an unexpected number of edges")
        self.topology = Topology.from_receiver_major(self.edge_index, torch.from_numpy(rowptr.astype(np.int32)), n_coarse, device,
                                                     int(degree.max()) if degree.size else 0)
        orig = self.first.to(torch.int32)[self.edge_index.to(device)]         # mesh node ids of both end points, [2, Ec]
        self.sender_orig, self.recv_orig = orig[0].contiguous(), orig[1].contiguous()

    @property
    def first_cpu(self) -> torch.Tensor:
        return self.first.cpu()
