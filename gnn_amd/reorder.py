"""Locality-preserving node numbering for arbitrary meshes.

Not in the reference (single device, PyG gathers do not care about ids).  The kernels and
the contiguous-range partitioner do: a workgroup updates the in-edges of a run of
consecutive receivers and gathers the projections of their senders, so neighbours that are
close in id are close in memory and hit in cache, and a node range owned by one rank has a
small halo only if ids follow the geometry.  ``node_order`` computes a permutation once per
mesh on the host; ``permute_frame`` applies it at the input edge and ``restore_nodes`` at
the output edge, so callers keep the reference's node order (core/rollout.py:36-43 returns
``pred[T, N, 3]`` in dataset order).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .graph import MeshGraph, NODE_FIELDS


def morton_order(mesh_pos: np.ndarray, bits: int = 16) -> np.ndarray:
    """Permutation (new position -> old id) that sorts the nodes along the Z-order curve of
    their reference coordinates (2-D; 16 bits per axis)."""
    pos = np.asarray(mesh_pos, dtype=np.float64)
    if pos.ndim != 2 or pos.shape[1] != 2:
        raise ValueError(f"morton_order: mesh_pos must be [N, 2], got {pos.shape}")
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    q = np.minimum(((pos - lo) / span * (2 ** bits)).astype(np.uint64), np.uint64(2 ** bits - 1))

    def spread(v):                                   # abcd -> 0a0b0c0d
        v = v & np.uint64(0xFFFFFFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
        return v

    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))
    return np.argsort(key, kind="stable").astype(np.int64)


def rcm_order(edge_index: np.ndarray, num_nodes: int) -> np.ndarray:
    """Reverse Cuthill-McKee permutation (new position -> old id) of the mesh graph: minimises
    the id distance between neighbours, which bounds both the gather footprint of a workgroup
    and the halo of a contiguous node range."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    ei = np.asarray(edge_index, dtype=np.int64)
    adj = csr_matrix((np.ones(ei.shape[1], dtype=np.int8), (ei[1], ei[0])), shape=(num_nodes, num_nodes))
    return np.asarray(reverse_cuthill_mckee(adj, symmetric_mode=True), dtype=np.int64)


def node_order(graph, method: str = "rcm") -> np.ndarray:
    """``method``: "rcm" (graph only), "morton" (coordinates only)."""
    if method == "rcm":
        return rcm_order(graph["edge_index"].detach().cpu().numpy(), graph["mesh_pos"].shape[0])
    if method == "morton":
        return morton_order(graph["mesh_pos"].detach().cpu().numpy())
    raise ValueError(f"node_order: unknown method {method!r} (rcm, morton)")


def bandwidth(edge_index: np.ndarray) -> int:
    """max |sender - receiver| over the edges (0 for an empty edge list)."""
    ei = np.asarray(edge_index)
    return int(np.abs(ei[0] - ei[1]).max()) if ei.shape[1] else 0


def inverse_permutation(order: np.ndarray) -> np.ndarray:
    inv = np.empty_like(order)
    inv[order] = np.arange(order.shape[0], dtype=order.dtype)
    return inv


def permute_frame(graph, order: np.ndarray, relabelled_edges: Optional[torch.Tensor] = None) -> MeshGraph:
    """The same frame with node ``order[i]`` moved to position i: node fields and ``target`` are
    reindexed, ``edge_index`` and ``cells`` are relabelled (their own order is kept).
    ``relabelled_edges`` lets a caller reuse one relabelled ``edge_index`` tensor for every frame
    of a trajectory, so the topology is built once."""
    n = graph["mesh_pos"].shape[0]
    if order.shape[0] != n:
        raise ValueError(f"permute_frame: permutation has {order.shape[0]} entries, mesh has {n} nodes")
    out = MeshGraph()
    inv = None
    for k, v in graph.items():
        if k in NODE_FIELDS:
            out[k] = v[torch.from_numpy(order).to(v.device)]
        elif k == "target" and torch.is_tensor(v):
            out[k] = v[:, torch.from_numpy(order).to(v.device)]
        elif k in ("edge_index", "cells") and torch.is_tensor(v):
            if k == "edge_index" and relabelled_edges is not None:
                out[k] = relabelled_edges
                continue
            if inv is None:
                inv = torch.from_numpy(inverse_permutation(order))
            out[k] = inv.to(v.device)[v]
        else:
            out[k] = v
    return out


def restore_nodes(t: torch.Tensor, order: np.ndarray, dim: int = 0) -> torch.Tensor:
    """Inverse of the node reindexing along ``dim``: ``restore_nodes(x[order]) == x``."""
    out = torch.empty_like(t)
    out.index_copy_(dim, torch.from_numpy(order).to(t.device), t)
    return out
