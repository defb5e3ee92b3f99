"""Graph container and receiver-major topology.

``MeshGraph`` is the minimal stand-in for the PyG ``Data`` frames the reference passes
around (run_train_multiscale.py:5,135; core/datasetclass.py:97-118): item access
(``g["phi"]``), attribute access (``g.phi``), ``.to(device)`` and ``deepcopy``.

``Topology`` is the receiver-major (CSR) view of ``edge_index`` the kernels consume.
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import numpy as np
import torch

from . import ops


class MeshGraph:
    def __init__(self, **fields):
        object.__setattr__(self, "_f", dict(fields))

    # item / attribute access -------------------------------------------------
    def __getitem__(self, key):
        return self._f[key]

    def __setitem__(self, key, value):
        self._f[key] = value

    def __contains__(self, key):
        return key in self._f

    def __getattr__(self, key):
        try:
            return object.__getattribute__(self, "_f")[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self._f[key] = value

    def keys(self):
        return self._f.keys()

    def items(self):
        return self._f.items()

    def to(self, device, non_blocking: bool = False):
        """A new frame whose tensors live on ``device`` (PyG semantics: not in place)."""
        out = MeshGraph()
        for k, v in self._f.items():
            out._f[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return out

    def __deepcopy__(self, memo):
        out = MeshGraph()
        for k, v in self._f.items():
            out._f[k] = v.clone() if torch.is_tensor(v) else copy.deepcopy(v, memo)
        return out

    def __repr__(self):
        parts = [f"{k}={tuple(v.shape) if torch.is_tensor(v) else v!r}" for k, v in self._f.items()]
        return "MeshGraph(" + ", ".join(parts) + ")"


NODE_FIELDS = ("mesh_pos", "world_pos", "phi", "swelling_phi", "node_type")


def batch_graphs(graphs) -> "MeshGraph":
    """Block-diagonal union of frames that share nothing: node fields are concatenated, the node
    ids in ``edge_index`` / ``cells`` of frame b are shifted by the node count of the frames before
    it.  This is how small meshes fill a GPU (several trajectories advance in one launch sequence);
    no message crosses blocks, so every block's result equals its stand-alone result as long as the
    normaliser statistics are the same.  ``mat_param`` is a per-frame constant broadcast to all
    nodes by the feature builder (core/meshgraphnet.py:219), so all frames must agree on it."""
    graphs = list(graphs)
    if not graphs:
        raise ValueError("batch_graphs: empty list")
    out = MeshGraph()
    first = graphs[0]
    offsets, n = [], 0
    for g in graphs:
        offsets.append(n)
        n += g["mesh_pos"].shape[0]
    for k in first.keys():
        v = first[k]
        if k in NODE_FIELDS:
            out[k] = torch.cat([g[k] for g in graphs], dim=0)
        elif k == "target":
            out[k] = torch.cat([g[k] for g in graphs], dim=1)
        elif k == "edge_index":
            out[k] = torch.cat([g[k] + o for g, o in zip(graphs, offsets)], dim=1)
        elif k == "cells":
            out[k] = torch.cat([g[k] + o for g, o in zip(graphs, offsets)], dim=0)
        elif k == "mat_param":
            for g in graphs[1:]:
                if not torch.equal(g[k], v):
                    raise ValueError("batch_graphs: frames with different mat_param cannot share a batch")
            out[k] = v
        else:
            out[k] = v
    out["batch_sizes"] = [g["mesh_pos"].shape[0] for g in graphs]
    return out


def unbatch_nodes(t: torch.Tensor, sizes, dim: int = 0):
    """Splits a node-indexed tensor of a batched frame back into its blocks."""
    return list(torch.split(t, list(sizes), dim=dim))


class Topology:
    """Receiver-major edge order for one ``edge_index``: in-edges of node n are CSR edges
    rowptr[n] .. rowptr[n+1]-1; ``perm[j]`` is the position of CSR edge j in ``edge_index``."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, device):
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise RuntimeError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
        self.num_nodes, self.num_edges = int(num_nodes), int(edge_index.shape[1])
        self.device = torch.device(device)
        if edge_index.is_cuda and edge_index.device == self._resolved(self.device):
            # the edge list is already in HBM: sort it there (range-checks every id)
            self.rowptr, self.sender, self.recv, self.perm = ops.receiver_csr_dev(edge_index.detach(), int(num_nodes))
            self.max_in_degree = int((self.rowptr[1:] - self.rowptr[:-1]).max()) if num_nodes > 0 and self.num_edges else 0
        else:
            ei = edge_index.detach().to("cpu", torch.int64).numpy()
            rowptr, cs, cr, perm = ops.receiver_csr_np(ei, int(num_nodes))   # range-checks every id
            self.rowptr = torch.from_numpy(rowptr).to(self.device)
            self.sender = torch.from_numpy(np.ascontiguousarray(cs)).to(self.device)
            self.recv = torch.from_numpy(np.ascontiguousarray(cr)).to(self.device)
            self.perm = torch.from_numpy(np.ascontiguousarray(perm)).to(self.device)
            self.max_in_degree = int(np.diff(rowptr).max()) if num_nodes > 0 else 0
        self._sender_view = None
        self._min_deg = {}

    @classmethod
    def from_receiver_major(cls, edge_index: torch.Tensor, rowptr: torch.Tensor, num_nodes: int, device, max_in_degree: int):
        """For an edge list that already is in receiver-major order (receivers ascending; ``rowptr`` int32 [N+1] given):
        no sort, no round trip to the host.  The k-nearest-neighbour graph of the coarse branch comes out this way."""
        t = cls.__new__(cls)
        t.num_nodes, t.num_edges, t.device = int(num_nodes), int(edge_index.shape[1]), torch.device(device)
        t.rowptr = rowptr.to(t.device, torch.int32).contiguous()
        t.sender = edge_index[0].to(t.device, torch.int32).contiguous()
        t.recv = edge_index[1].to(t.device, torch.int32).contiguous()
        t.perm = torch.arange(t.num_edges, dtype=torch.int64, device=t.device)
        t.max_in_degree = int(max_in_degree)
        t._sender_view, t._min_deg = None, {}
        return t

    def min_in_degree(self, first_rows: int = None) -> int:
        """Smallest in-degree among the first ``first_rows`` nodes (default: all), looked up once per topology."""
        n = self.num_nodes if first_rows is None else int(first_rows)
        if n not in self._min_deg:
            self._min_deg[n] = int((self.rowptr[1: n + 1] - self.rowptr[:n]).min()) if n > 0 else 0
        return self._min_deg[n]

    @staticmethod
    def _resolved(device: torch.device) -> torch.device:
        if device.type == "cuda" and device.index is None:
            return torch.device("cuda", torch.cuda.current_device())
        return device

    def sender_view(self):
        """(sender_rowptr int32 [N+1], sender_order int32 [E]): the CSR edges sent by node n are
        ``sender_order[sender_rowptr[n] : sender_rowptr[n+1]]`` in ascending CSR position.  Used
        by the backward pass to sum edge gradients per sender without atomics."""
        if self._sender_view is None:
            if self.sender.is_cuda:
                # the same stable device sort with the roles swapped: "receivers" = the CSR senders
                swapped = torch.stack([self.recv, self.sender]).to(torch.int64)
                rowptr, _, _, perm = ops.receiver_csr_dev(swapped, self.num_nodes)
                self._sender_view = (rowptr, perm.to(torch.int32))
            else:
                snd = self.sender.cpu().numpy()
                order = np.argsort(snd, kind="stable").astype(np.int32)
                rowptr = np.zeros(self.num_nodes + 1, dtype=np.int64)
                np.cumsum(np.bincount(snd, minlength=self.num_nodes), out=rowptr[1:])
                self._sender_view = (torch.from_numpy(rowptr.astype(np.int32)).to(self.device),
                                     torch.from_numpy(order).to(self.device))
        return self._sender_view


class TopologyCache:
    """Topologies keyed by the identity and version of the ``edge_index`` tensor."""

    def __init__(self, max_entries: int = 8):
        self._d: Dict[tuple, Topology] = {}
        self._max = max_entries

    def get(self, edge_index: torch.Tensor, num_nodes: int, device) -> Topology:
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), str(edge_index.device),
               int(num_nodes), str(device))
        t = self._d.get(key)
        if t is None:
            if len(self._d) >= self._max:
                self._d.pop(next(iter(self._d)))
            t = Topology(edge_index, num_nodes, device)
            t._keepalive = edge_index      # the key holds a data_ptr: keep that storage alive
            self._d[key] = t
        return t
