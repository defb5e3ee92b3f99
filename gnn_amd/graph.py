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


class Topology:
    """Receiver-major edge order for one ``edge_index``: in-edges of node n are CSR edges
    rowptr[n] .. rowptr[n+1]-1; ``perm[j]`` is the position of CSR edge j in ``edge_index``."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, device):
        ei = edge_index.detach().to("cpu", torch.int64).numpy()
        if ei.ndim != 2 or ei.shape[0] != 2:
            raise RuntimeError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
        rowptr, cs, cr, perm = ops.receiver_csr_np(ei, int(num_nodes))   # range-checks every id
        self.num_nodes, self.num_edges = int(num_nodes), int(ei.shape[1])
        self.device = torch.device(device)
        self.rowptr = torch.from_numpy(rowptr).to(self.device)
        self.sender = torch.from_numpy(np.ascontiguousarray(cs)).to(self.device)
        self.recv = torch.from_numpy(np.ascontiguousarray(cr)).to(self.device)
        self.perm = torch.from_numpy(np.ascontiguousarray(perm)).to(self.device)
        self.max_in_degree = int(np.diff(rowptr).max()) if num_nodes > 0 else 0
        self._sender_view = None

    def sender_view(self):
        """(sender_rowptr int32 [N+1], sender_order int32 [E]): the CSR edges sent by node n are
        ``sender_order[sender_rowptr[n] : sender_rowptr[n+1]]`` in ascending CSR position.  Used
        by the backward pass to sum edge gradients per sender without atomics."""
        if self._sender_view is None:
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
