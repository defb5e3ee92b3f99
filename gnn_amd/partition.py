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
gloo in the CPU tests) does the transport; the rows are packed and unpacked by the library's
``mgn_halo_*`` kernels into buffers the plan owns, so an exchange allocates nothing.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .graph import MeshGraph, Topology


def node_bounds(num_nodes: int, world_size: int) -> List[int]:
    return [(r * num_nodes) // world_size for r in range(world_size + 1)]


class HaloPlan:
    """Everything rank ``rank`` needs to run its share of a partitioned mesh."""

    def __init__(self, edge_index: np.ndarray, num_nodes: int, world_size: int, rank: int, bounds=None):
        """From the global edge list (every rank scans all of it; use :meth:`from_local` when a rank can
        produce its own part of the mesh).  ``bounds``: the ascending node-range limits [world_size + 1] when the
        ranges are not the equal split of :func:`node_bounds`."""
        ei = np.asarray(edge_index, dtype=np.int64)
        bounds = node_bounds(num_nodes, world_size) if bounds is None else [int(b) for b in bounds]
        lo, hi = bounds[rank], bounds[rank + 1]
        snd, rcv = ei[0], ei[1]
        mine = (rcv >= lo) & (rcv < hi)
        leaving = (snd >= lo) & (snd < hi) & ~mine
        self._build(ei[:, mine], np.nonzero(mine)[0], ei[:, leaving], int(num_nodes), int(ei.shape[1]), world_size, rank,
                    bounds)

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

    def _build(self, in_edges, in_edge_ids, out_edges, num_nodes, num_edges, world_size, rank, bounds=None):
        self.world_size, self.rank, self.global_nodes, self.global_edges = world_size, rank, num_nodes, num_edges
        self.bounds = node_bounds(num_nodes, world_size) if bounds is None else list(bounds)
        if (len(self.bounds) != world_size + 1 or self.bounds[0] != 0 or self.bounds[-1] != num_nodes
                or any(a > b for a, b in zip(self.bounds, self.bounds[1:]))):
            raise ValueError("HaloPlan: bounds must ascend from 0 to the number of nodes, one range per rank")
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
        host CSR routine), uploads the send list and the add-back lists of the reverse exchange, and
        creates the buffers the exchanges reuse from step to step (nothing is allocated per step)."""
        self.topology = Topology(torch.from_numpy(self.local_edge_index), self.num_local, device)
        if self.send_local_ids.size and (self.send_local_ids.min() < 0 or self.send_local_ids.max() >= self.num_owned):
            raise ValueError("HaloPlan: a send index lies outside the owned rows")
        dev = self.topology.rowptr.device
        self.device = dev
        self.num_send = int(self.send_local_ids.shape[0])
        self.send_idx = torch.from_numpy(self.send_local_ids.astype(np.int32)).to(dev)
        # reverse exchange: positions of the returned rows grouped by the owned row they are added to, peers
        # ascending within a row (the returned buffer is ordered like the send list, i.e. by peer)
        order = np.argsort(self.send_local_ids, kind="stable")
        rows, counts = np.unique(self.send_local_ids, return_counts=True)
        rowptr = np.zeros(rows.shape[0] + 1, dtype=np.int32)
        np.cumsum(counts, out=rowptr[1:])
        self.add_rows = torch.from_numpy(rows.astype(np.int32)).to(dev)
        self.add_rowptr = torch.from_numpy(rowptr).to(dev)
        self.add_pos = torch.from_numpy(order.astype(np.int32)).to(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.world, self.phi = torch.empty(self.num_local, 2, **f32), torch.empty(self.num_local, 1, **f32)
        self._state_send, self._state_recv = torch.empty(self.num_send, 3, **f32), torch.empty(self.num_halo, 3, **f32)
        self._bufs = {}
        self._in_flight = None
        self.group = group
        self._ranges(self.topology.rowptr.cpu().numpy())
        return self

    def _buffer(self, kind: str, width: int) -> torch.Tensor:
        """Persistent [num_send, width] buffer: ``kind`` "send" (rows packed for the peers) or "ret"
        (rows the peers return in the reverse exchange)."""
        buf = self._bufs.get((kind, width))
        if buf is None:
            buf = self._bufs[(kind, width)] = torch.empty(self.num_send, width, dtype=torch.float32, device=self.device)
        return buf

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
        flagged = np.nonzero(needs_halo)[0]
        starts = np.concatenate([[0], flagged[:-1] + 1])   # maximal unflagged runs end at flagged nodes
        k = int(np.argmax(flagged - starts))               # first of the longest
        a, b = int(starts[k]), int(flagged[k])
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

    # The three primitives below are the whole transport; everything else packs and unpacks with the library's
    # kernels into buffers the plan owns.  (The in-process tests replace exactly these.)
    def _all_to_all(self, recv, send, recv_counts, send_counts, async_op):
        """Rows of ``send`` (grouped by peer, ``send_counts``) to the peers, theirs into ``recv`` (``recv_counts``).
        Returns None once ``recv`` is ordered behind the exchange on the current stream, or with ``async_op`` a
        handle whose ``wait()`` establishes that order."""
        if self._host_staged(recv):
            recv_c = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(recv_c, send.cpu(), output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                   group=self.group)

            class _Staged:
                def wait(self_inner):
                    recv.copy_(recv_c)

            if async_op:
                return _Staged()
            recv.copy_(recv_c)
            return None
        return dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                      group=self.group, async_op=async_op)

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

    def _settle(self):
        """The send / return buffers are reused: an exchange still in flight is completed before the next one packs."""
        pending, self._in_flight = self._in_flight, None
        if pending is not None:
            pending.wait()

    def exchange_rows(self, buf: torch.Tensor, async_op: bool = False):
        """``buf`` is [num_local, W]; rows [0, num_owned) are valid.  Fills the halo rows with
        the owners' values.  Returns ``buf``, or with ``async_op`` a handle whose ``wait()`` makes
        the current stream wait for the halo rows (None when there is nothing to wait for)."""
        if self.world_size == 1:
            return None if async_op else buf
        self._settle()
        if buf.is_cuda:
            send = self._buffer("send", buf.shape[1])
            ops.halo_pack(buf, self.send_idx, send)
        else:                                     # CPU tensors (host-side tests of the plan): torch indexing
            send = buf.index_select(0, self.send_idx.to(torch.int64))
        work = self._all_to_all(buf[self.num_owned:], send, self.recv_counts, self.send_counts, async_op)
        if not async_op:
            return buf
        if work is None:
            return None
        plan = self

        class _Pending:
            def wait(self_inner):
                if plan._in_flight is self_inner:
                    plan._in_flight = None
                    work.wait()

        self._in_flight = _Pending()
        return self._in_flight

    def return_rows(self, buf: torch.Tensor, async_op: bool = False):
        """The reverse of :meth:`exchange_rows`, for gradients: the halo rows of ``buf``
        [num_local, W] travel to their owners, which add them to their own rows (peer by peer in
        rank order, so the result does not depend on timing).  Returns ``buf``, or with
        ``async_op`` a handle whose ``wait()`` completes the addition (None when there is nothing
        to do)."""
        if self.world_size == 1:
            return None if async_op else buf
        self._settle()
        recv = self._buffer("ret", buf.shape[1]) if buf.is_cuda else buf.new_empty((self.num_send, buf.shape[1]))
        work = self._all_to_all(recv, buf[self.num_owned:], self.send_counts, self.recv_counts, async_op)
        plan = self

        class _Pending:
            def wait(self_inner):
                if plan._in_flight is not self_inner:
                    return
                plan._in_flight = None
                if work is not None:
                    work.wait()
                plan._add_returned(buf, recv)

        self._in_flight = _Pending()
        if async_op:
            return self._in_flight
        self._in_flight.wait()
        return buf

    def _add_returned(self, buf: torch.Tensor, recv: torch.Tensor) -> None:
        if buf.is_cuda:
            ops.halo_unpack_add(recv, self.add_rowptr, self.add_rows, self.add_pos, buf)
            return
        off = 0
        for q in range(self.world_size):          # ids are unique within one peer's list
            cnt = self.send_counts[q]
            if cnt:
                ids = self.send_idx[off: off + cnt].to(torch.int64)
                buf[ids] += recv[off: off + cnt]
                off += cnt

    def exchange_state(self, world_pos: torch.Tensor, phi: torch.Tensor):
        """The per-step state as local [owned | halo] arrays with the halo rows refreshed from their owners
        (3 floats per halo node).  ``world_pos`` / ``phi`` hold at least the owned rows; the returned tensors
        are buffers of the plan, overwritten by the next call."""
        if self.world_size == 1:
            return world_pos, phi
        if not world_pos.is_cuda:
            st = torch.cat([world_pos, phi], dim=1)
            if st.shape[0] == self.num_owned:
                st = torch.cat([st, st.new_empty(self.num_halo, 3)])
            self.exchange_rows(st)
            return st[:, :2].contiguous(), st[:, 2:].contiguous()
        self._settle()
        ops.halo_stage_state(world_pos, phi, self.num_owned, self.send_idx, self.world, self.phi, self._state_send)
        self._all_to_all(self._state_recv, self._state_send, self.recv_counts, self.send_counts, False)
        ops.halo_unpack_state(self._state_recv, self.num_owned, self.world, self.phi)
        return self.world, self.phi

    def feature_stats(self, node_norm, node_rows, edge_norm, edge_rows):
        """``[mean | std]`` of the node- and the edge-feature normaliser for this step (both accumulate first while
        their accumulation is live): the float64 partial sums of both go through ONE all-reduce."""
        fn, fe = node_rows.shape[1], edge_rows.shape[1]
        buf = self._bufs.get(("sums", fn, fe))
        if buf is None:
            buf = self._bufs[("sums", fn, fe)] = torch.zeros(2 * (fn + fe), dtype=torch.float64, device=node_rows.device)
        sn = node_norm.local_sums(node_rows, True, out=buf[: 2 * fn])
        se = edge_norm.local_sums(edge_rows, True, out=buf[2 * fn:])
        if sn is not None or se is not None:
            self.all_reduce_(buf)
        return node_norm.fold(sn, self.global_nodes), edge_norm.fold(se, self.global_edges)

    def sum_hook(self, kind: str):
        """Hook for ``Normalizer.stats_for``: all-reduces the float64 partial sums and returns
        the global row count (``kind`` is "node" or "edge")."""
        total = self.global_nodes if kind == "node" else self.global_edges

        def hook(sums: torch.Tensor) -> int:
            self.all_reduce_(sums)
            return total
        return hook

    def _all_gather(self, parts, local_rows):
        """Third transport primitive: every rank's ``local_rows`` into ``parts`` (one tensor per rank)."""
        if self._host_staged(local_rows):
            host = [p.cpu() for p in parts]
            dist.all_gather(host, local_rows.cpu(), group=self.group)
            for p, h in zip(parts, host):
                p.copy_(h)
        else:
            dist.all_gather(parts, local_rows, group=self.group)

    def gather_owned(self, local_rows: torch.Tensor) -> torch.Tensor:
        """All ranks' owned rows concatenated in global node order (outputs, the coarse graph of a partitioned mesh).
        ``local_rows`` is [num_owned, ...]."""
        if self.world_size == 1:
            return local_rows
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world_size)]
        parts = [local_rows.new_empty((s,) + tuple(local_rows.shape[1:])) for s in sizes]
        self._all_gather(parts, local_rows.contiguous())
        return torch.cat(parts, dim=0)


class CoarsePartition:
    """The coarse graph of the multiscale option (core/meshgraphnet.py:237-283) on a partitioned mesh.

    Sampling is a sequential walk over the whole mesh, so every rank gathers the mesh coordinates once (8 bytes per
    node) and runs the same sampling and nearest-neighbour kernels: all ranks hold the same coarse graph.  A coarse
    node belongs to the rank that owns its mesh node; renumbered by ascending mesh node id the coarse nodes of a rank
    are one contiguous range, and the coarse blocks run on a second :class:`HaloPlan` over those ranges (the
    neighbours of a coarse node may lie anywhere, so this plan's halo is its own; renumbering keeps the order of a
    receiver's in-edges, hence the order of every sum).  Per step the coarse end points' state travels through that
    plan like the fine state does."""

    def __init__(self, plan: HaloPlan, mesh_local: torch.Tensor, sample_ratio: float, start: int):
        from .coarse import sample_and_connect
        dev = mesh_local.device
        mesh_global = plan.gather_owned(mesh_local[: plan.num_owned].to(torch.float32).contiguous())
        first, ei, counts = sample_and_connect(mesh_global, sample_ratio, [int(start)])
        m = counts[0]
        first_sorted, order = torch.sort(first)                   # new coarse id -> mesh node / id in sampling order
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(m, device=order.device)
        bounds_c = torch.searchsorted(first_sorted, torch.tensor(plan.bounds, dtype=torch.int64, device=first.device)).tolist()
        self.start, self.num_global, self.edges_global = int(start), m, int(ei.shape[1])
        self.plan = plan.__class__(new_id[ei].cpu().numpy(), m, plan.world_size, plan.rank, bounds=bounds_c).to(dev, plan.group)
        lo_c, hi_c = bounds_c[plan.rank], bounds_c[plan.rank + 1]
        # rows of the rank's fine arrays that are coarse nodes, in coarse order (int64 for row gather / scatter)
        self.first_local = (first_sorted[lo_c:hi_c] - plan.bounds[plan.rank]).to(dev).contiguous()
        ids = torch.from_numpy(self.plan.local_node_ids).to(first_sorted.device)
        self.mesh = mesh_global[first_sorted[ids]].to(dev).contiguous()        # coordinates of the local coarse rows
        self.topology = self.plan.topology
        self.num_nodes, self.num_owned = self.plan.num_local, self.plan.num_owned

    def state(self, world: torch.Tensor, phi: torch.Tensor):
        """(world_pos, phi) of the local coarse rows [owned | halo] from the rank's fine arrays."""
        return self.plan.exchange_state(world[self.first_local], phi[self.first_local])


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
