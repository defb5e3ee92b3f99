"""Mesh partitioning across GPUs: contiguous node ranges + one-hop halo exchange.

Not in the reference (it is single-device).  Rank r owns the nodes
``[bounds[r], bounds[r+1])`` and every edge whose *receiver* it owns, so the
receiver-segmented sum and the node update are purely local; the edge update needs the
sender projection of nodes owned by other ranks -- the halo.  Local node numbering is
``[owned (global order) | halo (global order)]``; because ranges are contiguous and
ascending, the halo block is already grouped by owner rank, so an
``all_to_all_single`` receives straight into it.

Per message-passing step one exchange moves the 128-float sender projections of the
halo nodes; per rollout step one exchange moves their (world_pos, phi); the normaliser
sums are all-reduced while accumulation is live.  ``torch.distributed`` (RCCL on GPUs,
gloo in the CPU tests) does the transport; packing uses ``index_select`` (a copy).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .graph import MeshGraph, Topology


def node_bounds(num_nodes: int, world_size: int) -> List[int]:
    return [(r * num_nodes) // world_size for r in range(world_size + 1)]


class HaloPlan:
    """Everything rank ``rank`` needs to run its share of a partitioned mesh."""

    def __init__(self, edge_index: np.ndarray, num_nodes: int, world_size: int, rank: int):
        """From the global edge list (every rank scans all of it; use :meth:`from_local` when a rank can
        produce its own part of the mesh)."""
        ei = np.asarray(edge_index, dtype=np.int64)
        bounds = node_bounds(num_nodes, world_size)
        lo, hi = bounds[rank], bounds[rank + 1]
        snd, rcv = ei[0], ei[1]
        mine = (rcv >= lo) & (rcv < hi)
        leaving = (snd >= lo) & (snd < hi) & ~mine
        self._build(ei[:, mine], np.nonzero(mine)[0], ei[:, leaving], int(num_nodes), int(ei.shape[1]), world_size, rank)

    @classmethod
    def from_local(cls, in_edges: np.ndarray, out_edges: np.ndarray, num_nodes: int, num_edges: int, world_size: int,
                   rank: int, in_edge_ids: Optional[np.ndarray] = None) -> "HaloPlan":
        """From the two edge sets a rank can list without seeing the whole mesh, in global node ids:
        ``in_edges`` [2, .] = every edge whose receiver the rank owns, in the order of the global edge
        list; ``out_edges`` [2, .] = every edge whose sender it owns and whose receiver it does not.
        ``num_nodes`` / ``num_edges`` are the global counts."""
        plan = cls.__new__(cls)
        plan._build(np.asarray(in_edges, dtype=np.int64),
                    None if in_edge_ids is None else np.asarray(in_edge_ids, dtype=np.int64),
                    np.asarray(out_edges, dtype=np.int64), int(num_nodes), int(num_edges), world_size, rank)
        return plan

    def _build(self, in_edges, in_edge_ids, out_edges, num_nodes, num_edges, world_size, rank):
        self.world_size, self.rank, self.global_nodes, self.global_edges = world_size, rank, num_nodes, num_edges
        self.bounds = node_bounds(num_nodes, world_size)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        owner_of = lambda ids: np.searchsorted(np.asarray(self.bounds[1:]), ids, side="right")
        ls, lr = in_edges[0], in_edges[1]
        if ls.size and (lr.min() < lo or lr.max() >= hi):
            raise ValueError("HaloPlan: an in-edge's receiver lies outside the rank's node range")
        self.edge_ids = in_edge_ids                               # global positions of the local edges (None if unknown)
        outside = (ls < lo) | (ls >= hi)
        self.halo_ids = np.unique(ls[outside])                    # global ids, ascending => grouped by owner
        self.num_owned, self.num_halo = hi - lo, int(self.halo_ids.shape[0])
        self.num_local = self.num_owned + self.num_halo
        # global -> local ids for the local edges
        loc_s = np.where(outside, self.num_owned + np.searchsorted(self.halo_ids, ls), ls - lo)
        self.local_edge_index = np.stack([loc_s, lr - lo]).astype(np.int64)
        self.local_node_ids = np.concatenate([np.arange(lo, hi, dtype=np.int64), self.halo_ids])
        halo_owner = owner_of(self.halo_ids)
        self.recv_counts = [int((halo_owner == q).sum()) for q in range(world_size)]
        # what every peer q needs from this rank: q's halo nodes that fall into [lo, hi)
        send_ids, self.send_counts = [], []
        cross_snd, cross_dst = out_edges[0], owner_of(out_edges[1])
        if cross_snd.size and (cross_snd.min() < lo or cross_snd.max() >= hi or (cross_dst == rank).any()):
            raise ValueError("HaloPlan: out-edges must lead from an owned sender to a receiver of another rank")
        for q in range(world_size):
            if q == rank:
                self.send_counts.append(0)
                continue
            need = np.unique(cross_snd[cross_dst == q])
            send_ids.append(need - lo)
            self.send_counts.append(int(need.shape[0]))
        self.send_local_ids = np.concatenate(send_ids) if send_ids else np.zeros(0, dtype=np.int64)
        self.topology: Optional[Topology] = None
        self.send_idx: Optional[torch.Tensor] = None
        self.group = None

    # ------------------------------------------------------------------ device side
    def to(self, device, group=None) -> "HaloPlan":
        """Builds the local receiver-major topology on ``device`` (needs libmgn_hip.so for its
        host CSR routine) and uploads the send list."""
        self.topology = Topology(torch.from_numpy(self.local_edge_index), self.num_local, device)
        self.send_idx = torch.from_numpy(self.send_local_ids).to(device)
        self.group = group
        self._ranges(self.topology.rowptr.cpu().numpy())
        return self

    def _ranges(self, rowptr: np.ndarray) -> None:
        """Splits the owned receivers into the longest contiguous run whose in-edges all come from
        owned senders (``interior``) and what is left on either side (``boundary``): the interior
        edges can be updated while the halo rows are still in flight.  Each range is
        (first node, end node, number of in-edges), ready for ``ops.mp_edge(node_range=...)``."""
        n_own = self.num_owned
        needs_halo = np.zeros(n_own + 1, dtype=bool)
        ls, lr = self.local_edge_index[0], self.local_edge_index[1]
        needs_halo[lr[ls >= n_own]] = True
        needs_halo[n_own] = True                       # sentinel
        best, start = (0, 0), 0
        for stop in np.nonzero(needs_halo)[0]:         # maximal unflagged runs end at flagged nodes
            if stop - start > best[1] - best[0]:
                best = (start, int(stop))
            start = int(stop) + 1
        a, b = best
        cnt = lambda n0, n1: int(rowptr[n1]) - int(rowptr[n0])
        self.interior = (a, b, cnt(a, b))
        self.boundary = [r for r in ((0, a, cnt(0, a)), (b, n_own, cnt(b, n_own))) if r[1] > r[0]]

    def local_graph(self, graph) -> MeshGraph:
        """The rank's slice of a full-mesh frame: per-node tensors restricted to
        [owned | halo] rows, ``edge_index`` in local ids."""
        ids = torch.from_numpy(self.local_node_ids)
        n = self.global_nodes
        out = MeshGraph()
        for k, v in graph.items():
            if k == "edge_index":
                out[k] = torch.from_numpy(self.local_edge_index)
            elif k == "target" and torch.is_tensor(v):
                out[k] = v[:, ids]
            elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n and k != "cells":
                out[k] = v[ids]
            else:
                out[k] = v
        return out

    def exchange_rows(self, buf: torch.Tensor, async_op: bool = False):
        """``buf`` is [num_local, W]; rows [0, num_owned) are valid.  Fills the halo rows with
        the owners' values.  Returns ``buf``, or with ``async_op`` a handle whose ``wait()`` makes
        the current stream wait for the halo rows (None when there is nothing to wait for)."""
        if self.world_size == 1:
            return None if async_op else buf
        send = buf.index_select(0, self.send_idx) if self.send_idx.numel() else buf.new_empty((0, buf.shape[1]))
        recv = buf[self.num_owned:]
        if self._host_staged(buf):
            return self._staged_all_to_all(recv, send, self.recv_counts, self.send_counts, async_op, buf)
        work = dist.all_to_all_single(recv, send, output_split_sizes=self.recv_counts, input_split_sizes=self.send_counts,
                                      group=self.group, async_op=async_op)
        return work if async_op else buf

    def return_rows(self, buf: torch.Tensor, async_op: bool = False):
        """The reverse of :meth:`exchange_rows`, for gradients: the halo rows of ``buf``
        [num_local, W] travel to their owners, which add them to their own rows (peer by peer in
        rank order, so the result does not depend on timing).  Returns ``buf``, or with
        ``async_op`` a handle whose ``wait()`` completes the addition (None when there is nothing
        to do)."""
        if self.world_size == 1:
            return None if async_op else buf
        send = buf[self.num_owned:].contiguous()
        recv = buf.new_empty((int(self.send_idx.numel()), buf.shape[1]))
        if self._host_staged(buf):
            work = self._staged_all_to_all(recv, send, self.send_counts, self.recv_counts, True, None)
        else:
            work = dist.all_to_all_single(recv, send, output_split_sizes=self.send_counts,
                                          input_split_sizes=self.recv_counts, group=self.group, async_op=async_op)
        plan = self

        class _Pending:
            def wait(self_inner):
                if work is not None:
                    work.wait()
                plan._add_returned(buf, recv)

        if async_op:
            return _Pending()
        _Pending().wait()                         # the host-staged transport hands back a handle in both modes
        return buf

    def _add_returned(self, buf: torch.Tensor, recv: torch.Tensor) -> None:
        off = 0
        for q in range(self.world_size):          # ids are unique within one peer's list
            cnt = self.send_counts[q]
            if cnt:
                ids = self.send_idx[off: off + cnt]
                buf[ids] += recv[off: off + cnt]
                off += cnt

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """Sum over the ranks of the plan's group, in place."""
        if self.world_size > 1:
            if self._host_staged(t):
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(c)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    # A process group without a device transport (gloo) moves device tensors through host memory.  This is
    # for rehearsing the multi-rank path on a machine without RCCL peers (several ranks on one GPU); the
    # arithmetic still runs in the HIP kernels.
    def _host_staged(self, t: torch.Tensor) -> bool:
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _staged_all_to_all(self, recv, send, out_splits, in_splits, async_op, ret):
        recv_c = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(recv_c, send.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=self.group)

        class _Done:
            def wait(self_inner):
                recv.copy_(recv_c)

        if async_op:
            return _Done()
        recv.copy_(recv_c)
        return ret

    def exchange_state(self, world_pos: torch.Tensor, phi: torch.Tensor):
        """Refreshes the halo rows of the per-step state (3 floats per halo node)."""
        if self.world_size == 1:
            return world_pos, phi
        st = torch.cat([world_pos, phi], dim=1).contiguous()
        self.exchange_rows(st)
        return st[:, :2].contiguous(), st[:, 2:].contiguous()

    def sum_hook(self, kind: str):
        """Hook for ``Normalizer.stats_for``: all-reduces the float64 partial sums and returns
        the global row count (``kind`` is "node" or "edge")."""
        total = self.global_nodes if kind == "node" else self.global_edges

        def hook(sums: torch.Tensor) -> int:
            self.all_reduce_(sums)
            return total
        return hook

    def gather_owned(self, local_rows: torch.Tensor) -> torch.Tensor:
        """All ranks' owned rows concatenated in global node order (for outputs / tests).
        ``local_rows`` is [num_owned, ...]."""
        if self.world_size == 1:
            return local_rows
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world_size)]
        parts = [local_rows.new_empty((s,) + tuple(local_rows.shape[1:])) for s in sizes]
        dist.all_gather(parts, local_rows.contiguous(), group=self.group)
        return torch.cat(parts, dim=0)


def all_reduce_gradients(model, group=None, average: bool = True) -> None:
    """Data-parallel training over replicas (each rank a different trajectory): sums -- or averages --
    the ``.grad`` of all parameters across the group with one all-reduce of one flat buffer."""
    params = [p for p in model.parameters() if p.grad is not None]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        p.grad.copy_(flat[off: off + p.numel()].view_as(p))
        off += p.numel()


def attach(model, plan: HaloPlan):
    """Switches ``model`` to partitioned execution on ``plan`` (``None`` switches back)."""
    model.__dict__["halo"] = plan
    model.__dict__["_ws"] = None
    return model
