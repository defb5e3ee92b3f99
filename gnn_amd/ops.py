"""Tensor-level wrappers over the C ABI (include/mgn_hip.h).

Every wrapper checks dtype, device, contiguity and shape on the host before the kernel
is launched, passes raw device pointers and the current HIP stream, and raises
RuntimeError on any non-zero return code.  torch is used for memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import H, MAX_DEC_OUT, MAX_FEAT, MgnMlp


# Optional per-kernel timing: when a list is installed, every `timed(name)` block appends
# (name, start_event, end_event) recorded on the current stream.
_event_log = None


def set_event_log(log):
    """Installs (or with None removes) the list that receives kernel timing events."""
    global _event_log
    _event_log = log


def timing_enabled() -> bool:
    return _event_log is not None


class timed:
    __slots__ = ("name", "start")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _event_log is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if _event_log is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            _event_log.append((self.name, self.start, end))
        return False


def _timed_op(fn):
    """Wraps a kernel wrapper so that its launches are logged under its own name while an event log is
    installed (no cost otherwise beyond one comparison)."""
    import functools
    name = fn.__name__

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if _event_log is None:
            return fn(*args, **kwargs)
        with timed(name):
            return fn(*args, **kwargs)
    return wrapper


def _stream() -> int:
    """Raw handle of torch's current HIP stream."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _bad_tensor(t, dtype, shape, name: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")
    raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def _dev(t: torch.Tensor, dtype, shape, name: str) -> int:
    """Device address of ``t`` after the host-side checks (this runs several times per launch: the
    checks are ordered so that the passing case costs about a microsecond)."""
    try:
        ok = t.is_cuda and t.dtype is dtype and t.is_contiguous()
    except AttributeError:
        ok = False
    if ok and shape is not None:
        ts = t.shape
        if len(ts) != len(shape):
            ok = False
        else:
            for d, s in zip(ts, shape):
                if s is not None and d != s:
                    ok = False
                    break
    if not ok:
        _bad_tensor(t, dtype, shape, name)
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], dtype, shape, name: str) -> int:
    return 0 if t is None else _dev(t, dtype, shape, name)


class MlpWeights:
    """Device pointers of one two-Linear MLP; keeps the tensors alive."""

    def __init__(self, w0, b0, w1, b1, ln_w=None, ln_b=None, in_dim=None, out_dim=H):
        self.tensors = (w0, b0, w1, b1, ln_w, ln_b)
        in_dim = w0.shape[1] if in_dim is None else in_dim
        self.in_dim, self.out_dim = int(in_dim), int(out_dim)
        self.c = MgnMlp(_dev(w0, torch.float32, (H, in_dim), "w0"), _dev(b0, torch.float32, (H,), "b0"),
                        _dev(w1, torch.float32, (out_dim, H), "w1"), _dev(b1, torch.float32, (out_dim,), "b1"),
                        _opt(ln_w, torch.float32, (H,), "ln_w"), _opt(ln_b, torch.float32, (H,), "ln_b"))

    @staticmethod
    def from_sequential(seq, ln_index: Optional[int]) -> "MlpWeights":
        """seq[0], seq[2] are the Linears; seq[ln_index] the LayerNorm (None if absent)."""
        ln = seq[ln_index] if ln_index is not None else None
        return MlpWeights(seq[0].weight.data, seq[0].bias.data, seq[2].weight.data, seq[2].bias.data,
                          None if ln is None else ln.weight.data, None if ln is None else ln.bias.data,
                          in_dim=seq[0].weight.shape[1], out_dim=seq[2].weight.shape[0])

    def ref(self):
        return C.byref(self.c)


# --------------------------------------------------------------------------- host: edge build
def cells_to_edge_index_np(cells: np.ndarray) -> Tuple[np.ndarray, int]:
    cells = np.ascontiguousarray(np.asarray(cells, dtype=np.int64).reshape(-1, 3))
    nc = cells.shape[0]
    cap = max(6 * nc, 1)
    snd = np.empty(cap, dtype=np.int64)
    rcv = np.empty(cap, dtype=np.int64)
    ne, nn = C.c_int64(0), C.c_int64(0)
    rc = _lib.lib().mgn_cells_to_edge_index(cells.ctypes.data, nc, snd.ctypes.data, rcv.ctypes.data, cap,
                                            C.addressof(ne), C.addressof(nn))
    _lib.check(rc, "mgn_cells_to_edge_index")
    return np.stack([snd[:ne.value], rcv[:ne.value]]), int(nn.value)


def receiver_csr_np(edge_index: np.ndarray, num_nodes: int):
    ei = np.ascontiguousarray(np.asarray(edge_index, dtype=np.int64))
    if ei.ndim != 2 or ei.shape[0] != 2:
        raise RuntimeError(f"edge_index must be [2, E], got {ei.shape}")
    ne = ei.shape[1]
    snd, rcv = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    rowptr = np.empty(num_nodes + 1, dtype=np.int32)
    cs = np.empty(max(ne, 1), dtype=np.int32)
    cr = np.empty(max(ne, 1), dtype=np.int32)
    perm = np.empty(max(ne, 1), dtype=np.int64)
    rc = _lib.lib().mgn_receiver_csr(snd.ctypes.data, rcv.ctypes.data, ne, num_nodes, rowptr.ctypes.data,
                                     cs.ctypes.data, cr.ctypes.data, perm.ctypes.data)
    _lib.check(rc, "mgn_receiver_csr")
    return rowptr, cs[:ne], cr[:ne], perm[:ne]


# --------------------------------------------------------------------------- device: edge build
def cells_to_edge_index_dev(cells: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """Device version of :func:`cells_to_edge_index_np` for a CUDA ``cells`` tensor: returns
    (edge_index int64 [2, E] on the same device, number of nodes = max id + 1)."""
    cells = cells.reshape(-1, 3).to(torch.int64).contiguous()
    nc = cells.shape[0]
    dev = cells.device
    if nc == 0:
        return torch.empty(2, 0, dtype=torch.int64, device=dev), 0
    with torch.cuda.device(dev):
        need = int(_lib.lib().mgn_edge_build_workspace_bytes(nc))
        if need < 0:
            raise RuntimeError("cells_to_edge_index: too many cells for the device build (6 * cells must stay below 2^31)")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(2, 6 * nc, dtype=torch.int64, device=dev)
        ne, nn = C.c_int64(0), C.c_int64(0)
        rc = _lib.lib().mgn_cells_to_edge_index_dev(_dev(cells, torch.int64, (nc, 3), "cells"), nc, ws.data_ptr(), need,
                                                    out[0].data_ptr(), out[1].data_ptr(), 6 * nc, C.addressof(ne),
                                                    C.addressof(nn), _stream())
        _lib.check(rc, "mgn_cells_to_edge_index_dev")
    return out[:, : ne.value].contiguous(), int(nn.value)


def receiver_csr_dev(edge_index: torch.Tensor, num_nodes: int):
    """Device version of :func:`receiver_csr_np` for a CUDA ``edge_index``: (rowptr int32 [N+1],
    csr_sender int32 [E], csr_recv int32 [E], perm int64 [E]) on the same device."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise RuntimeError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    ei = edge_index.to(torch.int64).contiguous()
    ne, dev = ei.shape[1], ei.device
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    cs = torch.empty(ne, dtype=torch.int32, device=dev)
    cr = torch.empty(ne, dtype=torch.int32, device=dev)
    perm = torch.empty(ne, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        need = int(_lib.lib().mgn_receiver_csr_workspace_bytes(ne))
        if need < 0:
            raise RuntimeError("receiver_csr: sizes exceed the int32 index range")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        rc = _lib.lib().mgn_receiver_csr_dev(ei[0].data_ptr(), ei[1].data_ptr(), ne, int(num_nodes), ws.data_ptr(), need,
                                             _dev(rowptr, torch.int32, (num_nodes + 1,), "rowptr"), cs.data_ptr(),
                                             cr.data_ptr(), perm.data_ptr(), _stream())
        _lib.check(rc, "mgn_receiver_csr_dev")
    return rowptr, cs, cr, perm


# --------------------------------------------------------------------------- device: processor
def mp_project(x, w0, b0, ps, pr):
    n = x.shape[0]
    rc = _lib.lib().mgn_mp_project(_dev(x, torch.float32, (n, H), "x"), _dev(w0, torch.float32, (H, 3 * H), "edge w0"),
                                   _dev(b0, torch.float32, (H,), "edge b0"), _dev(ps, torch.float32, (n, H), "ps"),
                                   _dev(pr, torch.float32, (n, H), "pr"), n, _stream())
    _lib.check(rc, "mgn_mp_project")


def mp_edge(e_in, ps, pr, rowptr, csr_sender, csr_recv, mlp: MlpWeights, e_out, aggr, node_range=None, edge_weight=None,
            packed=None):
    """``node_range = (n0, n1, edge_count)`` restricts the launch to the receivers n0 .. n1-1 and
    their ``edge_count`` in-edges (= rowptr[n1] - rowptr[n0], known to the caller on the host).
    ``edge_weight`` [E] (CSR order): weights of the messages in the receivers' sums (attention).
    ``packed``: device addresses (w0[:, 2H:], w1) of the sets prepared by :class:`PackedWeights`."""
    ne, n = e_in.shape[0], ps.shape[0]
    if mlp.in_dim != 3 * H or mlp.out_dim != H:
        raise RuntimeError("mp_edge: edge MLP must be 3H -> H -> H")
    if n * H >= 2 ** 31:
        raise RuntimeError("mp_edge: node rows are addressed with 32-bit element offsets (at most 16 777 215 nodes per GPU)")
    rp = _dev(rowptr, torch.int32, (None,), "rowptr")
    if rowptr.shape[0] < aggr.shape[0] + 1:
        raise RuntimeError("mp_edge: rowptr is shorter than the number of receivers + 1")
    n_sub, e_sub = aggr.shape[0], ne
    if node_range is not None:
        n0, n1, e_sub = (int(v) for v in node_range)
        if not (0 <= n0 <= n1 <= aggr.shape[0]) or not (0 <= e_sub <= ne):
            raise RuntimeError(f"mp_edge: bad receiver sub-range {node_range}")
        rp = C.c_void_p(rowptr.data_ptr() + 4 * n0)
        n_sub = n1 - n0
        if n_sub == 0 or e_sub == 0:
            return
    if packed is not None:
        rc = _lib.lib().mgn_mp_edge_packed(
            _dev(e_in, torch.float32, (ne, H), "e_in"), _dev(ps, torch.float32, (n, H), "ps"),
            _dev(pr, torch.float32, (n, H), "pr"), rp, _dev(csr_sender, torch.int32, (ne,), "csr_sender"),
            _dev(csr_recv, torch.int32, (ne,), "csr_recv"), mlp.ref(), packed[0], packed[1],
            _opt(edge_weight, torch.float32, (ne,), "edge_weight"),
            _opt(e_out, torch.float32, (ne, H), "e_out"), _dev(aggr, torch.float32, (None, H), "aggr"), e_sub, n_sub, _stream())
        _lib.check(rc, "mgn_mp_edge_packed")
        return
    if edge_weight is not None:
        rc = _lib.lib().mgn_mp_edge_weighted(
            _dev(e_in, torch.float32, (ne, H), "e_in"), _dev(ps, torch.float32, (n, H), "ps"),
            _dev(pr, torch.float32, (n, H), "pr"), rp, _dev(csr_sender, torch.int32, (ne,), "csr_sender"),
            _dev(csr_recv, torch.int32, (ne,), "csr_recv"), mlp.ref(), _dev(edge_weight, torch.float32, (ne,), "edge_weight"),
            _opt(e_out, torch.float32, (ne, H), "e_out"), _dev(aggr, torch.float32, (None, H), "aggr"), e_sub, n_sub, _stream())
        _lib.check(rc, "mgn_mp_edge_weighted")
        return
    rc = _lib.lib().mgn_mp_edge(_dev(e_in, torch.float32, (ne, H), "e_in"), _dev(ps, torch.float32, (n, H), "ps"),
                                _dev(pr, torch.float32, (n, H), "pr"), rp,
                                _dev(csr_sender, torch.int32, (ne,), "csr_sender"),
                                _dev(csr_recv, torch.int32, (ne,), "csr_recv"), mlp.ref(),
                                _opt(e_out, torch.float32, (ne, H), "e_out"), _dev(aggr, torch.float32, (None, H), "aggr"),
                                e_sub, n_sub, _stream())
    _lib.check(rc, "mgn_mp_edge")


def edge_attention(a_proj, rowptr, csr_sender, csr_recv, alpha, edge_weight):
    """Message weights of the attention option from the projected node latents ``a_proj`` [N,H]."""
    n, ne = a_proj.shape[0], csr_sender.shape[0]
    rc = _lib.lib().mgn_edge_attention(_dev(a_proj, torch.float32, (n, H), "a_proj"), _dev(rowptr, torch.int32, (n + 1,), "rowptr"),
                                       _dev(csr_sender, torch.int32, (ne,), "csr_sender"), _dev(csr_recv, torch.int32, (ne,), "csr_recv"),
                                       _dev(alpha, torch.float32, (ne,), "alpha"), _dev(edge_weight, torch.float32, (ne,), "edge_weight"),
                                       ne, n, _stream())
    _lib.check(rc, "mgn_edge_attention")


def mp_node(x, aggr, rowptr, mlp: MlpWeights, x_out):
    n = x.shape[0]
    if mlp.in_dim != 2 * H or mlp.out_dim != H:
        raise RuntimeError("mp_node: node MLP must be 2H -> H -> H")
    rc = _lib.lib().mgn_mp_node(_dev(x, torch.float32, (n, H), "x"), _dev(aggr, torch.float32, (n, H), "aggr"),
                                _dev(rowptr, torch.int32, (n + 1,), "rowptr"), mlp.ref(),
                                _dev(x_out, torch.float32, (n, H), "x_out"), n, _stream())
    _lib.check(rc, "mgn_mp_node")


def mp_node_project(x, aggr, rowptr, mlp: MlpWeights, next_w0, next_b0, x_out, ps, pr):
    n = x.shape[0]
    if mlp.in_dim != 2 * H or mlp.out_dim != H:
        raise RuntimeError("mp_node_project: node MLP must be 2H -> H -> H")
    rc = _lib.lib().mgn_mp_node_project(
        _dev(x, torch.float32, (n, H), "x"), _dev(aggr, torch.float32, (n, H), "aggr"),
        _dev(rowptr, torch.int32, (n + 1,), "rowptr"), mlp.ref(), _dev(next_w0, torch.float32, (H, 3 * H), "next edge w0"),
        _dev(next_b0, torch.float32, (H,), "next edge b0"), _dev(x_out, torch.float32, (n, H), "x_out"),
        _dev(ps, torch.float32, (n, H), "ps"), _dev(pr, torch.float32, (n, H), "pr"), n, _stream())
    _lib.check(rc, "mgn_mp_node_project")


def mp_node_packed(x, aggr, rowptr, mlp: MlpWeights, packed, x_out, next_w0=None, next_b0=None, ps=None, pr=None):
    """:func:`mp_node` (``next_w0`` None) or :func:`mp_node_project` with prepared weight sets: ``packed`` = device
    addresses (w0, w1, next ws, next wr) from :class:`PackedWeights`, the last two 0 without a next layer."""
    n = x.shape[0]
    if mlp.in_dim != 2 * H or mlp.out_dim != H:
        raise RuntimeError("mp_node_packed: node MLP must be 2H -> H -> H")
    proj = next_w0 is not None
    rc = _lib.lib().mgn_mp_node_packed(
        _dev(x, torch.float32, (n, H), "x"), _dev(aggr, torch.float32, (n, H), "aggr"),
        _dev(rowptr, torch.int32, (n + 1,), "rowptr"), mlp.ref(), packed[0], packed[1],
        _dev(next_w0, torch.float32, (H, 3 * H), "next edge w0") if proj else None,
        _dev(next_b0, torch.float32, (H,), "next edge b0") if proj else None,
        packed[2] if proj else None, packed[3] if proj else None, _dev(x_out, torch.float32, (n, H), "x_out"),
        _dev(ps, torch.float32, (n, H), "ps") if proj else None, _dev(pr, torch.float32, (n, H), "pr") if proj else None,
        n, _stream())
    _lib.check(rc, "mgn_mp_node_packed")


class PackedWeights:
    """Weight sets of a stack of processor blocks in the layout the fp16-split kernels load fully coalesced
    (``mgn_pack_weights``): per block the edge MLP's ``w0[:, 2H:]`` and ``w1``, the node MLP's ``w0`` and ``w1``, and the
    sender / receiver thirds of the NEXT block's edge ``w0``.  ``refresh()`` converts all of them in one launch -- a
    snapshot of the parameters, so it runs at the start of every call that uses the sets; the descriptor table and the
    buffer are rebuilt only when a parameter tensor moves."""

    def __init__(self):
        self.key = None

    def refresh(self, blocks):
        params = []
        for b in blocks:
            em, nm = b.edge_mlp._modules, b.node_mlp._modules
            params += [em["0"].weight.data, em["2"].weight.data, nm["0"].weight.data, nm["2"].weight.data]
        key = tuple(p.data_ptr() for p in params)
        if key != self.key:
            nb = len(blocks)
            sl, sizes = [], []
            for i in range(nb):
                e0, e1, n0, n1 = params[4 * i: 4 * i + 4]
                for t in (e0, e1, n0, n1):
                    _dev(t, torch.float32, (H, None), "processor weight")
                sets = [(e0, 2 * H, H), (e1, 0, H), (n0, 0, 2 * H), (n1, 0, H)]
                if i + 1 < nb:
                    nxt = params[4 * (i + 1)]
                    sets += [(nxt, 0, H), (nxt, H, H)]
                for t, col0, k in sets:
                    sl.append(_lib.MgnWeightSlice(t.data_ptr(), t.shape[1], col0, k, 0))
                    sizes.append(int(_lib.lib().mgn_packed_weight_bytes(k)))
            self.count = len(sl)
            self.buf = torch.empty(sum(sizes), dtype=torch.uint8, device=params[0].device)
            base, offs = self.buf.data_ptr(), [0]
            for sz in sizes:
                offs.append(offs[-1] + sz)
            # one mgn_pack_weights call takes at most MAX_PACK_SLICES sets: (table, count, byte offset of its first set)
            cap = _lib.MAX_PACK_SLICES
            self.calls = [((_lib.MgnWeightSlice * len(sl[i: i + cap]))(*sl[i: i + cap]), len(sl[i: i + cap]), offs[i])
                          for i in range(0, len(sl), cap)]
            self.edge, self.node = [], []
            j = 0
            for i in range(nb):
                a = [C.c_void_p(base + offs[j + q]) for q in range(4)]
                nx = [C.c_void_p(base + offs[j + 4]), C.c_void_p(base + offs[j + 5])] if i + 1 < nb else [None, None]
                self.edge.append((a[0], a[1]))
                self.node.append((a[2], a[3], nx[0], nx[1]))
                j += 6 if i + 1 < nb else 4
            self.keep = params                      # the table holds raw addresses of these tensors
            self.key = key
        for table, count, off in self.calls:
            rc = _lib.lib().mgn_pack_weights(table, count, C.c_void_p(self.buf.data_ptr() + off), _stream())
            _lib.check(rc, "mgn_pack_weights")
        return self


# --------------------------------------------------------------------------- device: features etc.
def _sums_scratch(partials, rows: int, n_feat: int) -> int:
    p = _dev(partials, torch.float64, (None,), "partials")
    if partials.numel() * 8 < feature_sums_scratch_bytes(rows, n_feat):
        raise RuntimeError("partial-sum buffer is smaller than mgn_feature_sums_scratch_bytes()")
    return p


def node_features(world_pos, mesh_pos, phi, swelling_phi, node_type, mat_param, feat, partials=None):
    """``partials`` (float64 buffer of ``feature_sums_scratch_bytes`` bytes): the same launch also forms the per-block
    column sums of the rows for :func:`norm_fold` (``mgn_node_features_sums``)."""
    n = world_pos.shape[0]
    ns = swelling_phi.shape[1]
    f = 3 + ns + 5 + (2 if mat_param is not None else 0)
    args = (_dev(world_pos, torch.float32, (n, 2), "world_pos"), _dev(mesh_pos, torch.float32, (n, 2), "mesh_pos"),
            _dev(phi, torch.float32, (n, 1), "phi"), _dev(swelling_phi, torch.float32, (n, ns), "swelling_phi"), ns,
            _dev(node_type, torch.float32, (n, 5), "node_type"), _opt(mat_param, torch.float32, (2,), "mat_param"),
            _dev(feat, torch.float32, (n, f), "feat"), n)
    if partials is not None and n > 0:
        _lib.check(_lib.lib().mgn_node_features_sums(*args, _sums_scratch(partials, n, f), _stream()), "mgn_node_features_sums")
    else:
        _lib.check(_lib.lib().mgn_node_features(*args, _stream()), "mgn_node_features")


def edge_features(world_pos, mesh_pos, phi, csr_sender, csr_recv, feat, partials=None):
    """``partials``: as in :func:`node_features` (``mgn_edge_features_sums``)."""
    n, ne = world_pos.shape[0], csr_sender.shape[0]
    args = (_dev(world_pos, torch.float32, (n, 2), "world_pos"), _dev(mesh_pos, torch.float32, (n, 2), "mesh_pos"),
            _dev(phi, torch.float32, (n, 1), "phi"), _dev(csr_sender, torch.int32, (ne,), "csr_sender"),
            _dev(csr_recv, torch.int32, (ne,), "csr_recv"), _dev(feat, torch.float32, (ne, 7), "feat"), ne)
    if partials is not None and ne > 0:
        _lib.check(_lib.lib().mgn_edge_features_sums(*args, _sums_scratch(partials, ne, 7), _stream()), "mgn_edge_features_sums")
    else:
        _lib.check(_lib.lib().mgn_edge_features(*args, _stream()), "mgn_edge_features")


def norm_fold(partials, rows: int, acc_sum, acc_sumsq, count_dev, count: int, eps: float, mean_std):
    """Adds the per-block partial sums of ``rows`` rows, folds the totals into the accumulators and writes [mean | std]
    (``mgn_norm_fold``); ``count``: the row count after this call, or ``count_dev`` (int64 device word, advanced by
    ``rows``) for captured launch sequences."""
    f = acc_sum.numel()
    rc = _lib.lib().mgn_norm_fold(_sums_scratch(partials, rows, f), rows, f, _dev(acc_sum, torch.float32, None, "acc_sum"),
                                  _dev(acc_sumsq, torch.float32, None, "acc_sumsq"), _opt(count_dev, torch.int64, (1,), "count_dev"),
                                  int(count), float(eps), _dev(mean_std, torch.float32, (2 * f,), "mean_std"), _stream())
    _lib.check(rc, "mgn_norm_fold")


def feature_sums_scratch_bytes(rows: int, n_feat: int) -> int:
    return int(_lib.lib().mgn_feature_sums_scratch_bytes(rows, n_feat))


def feature_sums(feat, sums, scratch):
    m, f = feat.shape
    need = feature_sums_scratch_bytes(m, f)
    if scratch.numel() * scratch.element_size() < need:
        raise RuntimeError(f"feature_sums: scratch holds {scratch.numel() * scratch.element_size()} B, needs {need}")
    rc = _lib.lib().mgn_feature_sums(_dev(feat, torch.float32, (m, f), "feat"), m, f,
                                     _dev(sums, torch.float64, (2 * f,), "sums"),
                                     _dev(scratch, scratch.dtype, None, "scratch"), _stream())
    _lib.check(rc, "mgn_feature_sums")


def feature_sums_static_prepare():
    """Resolves, for the current device, the address of the buffer :func:`feature_sums_static` uses (needed once per
    device before that function is called under HIP graph capture)."""
    _lib.check(_lib.lib().mgn_feature_sums_static_prepare(), "mgn_feature_sums_static_prepare")


def feature_sums_static(feat, sums):
    """:func:`feature_sums` through the entry point that allocates nothing and takes no workspace (the
    partial sums live in a buffer inside the library, one per GPU; calls overlapping on one GPU race)."""
    m, f = feat.shape
    rc = _lib.lib().mgn_feature_sums_static(_dev(feat, torch.float32, (m, f), "feat"), m, f,
                                            _dev(sums, torch.float64, (2 * f,), "sums"), _stream())
    _lib.check(rc, "mgn_feature_sums_static")


def norm_update(acc_sum, acc_sumsq, sums, count: int, eps: float, mean_std):
    f = acc_sum.numel()
    rc = _lib.lib().mgn_norm_update(_dev(acc_sum, torch.float32, None, "acc_sum"),
                                    _dev(acc_sumsq, torch.float32, None, "acc_sumsq"),
                                    _opt(sums, torch.float64, (2 * f,), "sums"), f, int(count), float(eps),
                                    _dev(mean_std, torch.float32, (2 * f,), "mean_std"), _stream())
    _lib.check(rc, "mgn_norm_update")


def norm_update_counted(acc_sum, acc_sumsq, sums, count_dev, add_rows: int, eps: float, mean_std):
    f = acc_sum.numel()
    rc = _lib.lib().mgn_norm_update_counted(_dev(acc_sum, torch.float32, None, "acc_sum"),
                                            _dev(acc_sumsq, torch.float32, None, "acc_sumsq"),
                                            _opt(sums, torch.float64, (2 * f,), "sums"), f,
                                            _dev(count_dev, torch.int64, (1,), "count_dev"), int(add_rows), float(eps),
                                            _dev(mean_std, torch.float32, (2 * f,), "mean_std"), _stream())
    _lib.check(rc, "mgn_norm_update_counted")


def normalize(x2d, mean_std, out, inverse: bool = False):
    m, f = x2d.shape
    rc = _lib.lib().mgn_normalize(_dev(x2d, torch.float32, (m, f), "x"), _dev(mean_std, torch.float32, (2 * f,), "mean_std"),
                                  _dev(out, torch.float32, (m, f), "out"), m, f, 1 if inverse else 0, _stream())
    _lib.check(rc, "mgn_normalize")


def inject_noise(world_pos, phi, node_type, z_world, z_phi, level, conn_length, phi_range, out_world, out_phi):
    n = world_pos.shape[0]
    rc = _lib.lib().mgn_inject_noise(_dev(world_pos, torch.float32, (n, 2), "world_pos"), _dev(phi, torch.float32, (n, 1), "phi"),
                                     _dev(node_type, torch.float32, (n, 5), "node_type"),
                                     _dev(z_world, torch.float32, (n, 2), "z_world"), _dev(z_phi, torch.float32, (n, 1), "z_phi"),
                                     float(level), float(conn_length), float(phi_range),
                                     _dev(out_world, torch.float32, (n, 2), "out_world"),
                                     _dev(out_phi, torch.float32, (n, 1), "out_phi"), n, _stream())
    _lib.check(rc, "mgn_inject_noise")


def encode(feat, mean_std, mlp: MlpWeights, out):
    m, f = feat.shape
    if f > MAX_FEAT or mlp.in_dim != f or mlp.out_dim != H:
        raise RuntimeError(f"encode: feature width {f} (max {MAX_FEAT}) does not match the encoder ({mlp.in_dim} -> {mlp.out_dim})")
    rc = _lib.lib().mgn_encode(_dev(feat, torch.float32, (m, f), "feat"), f, _opt(mean_std, torch.float32, (2 * f,), "mean_std"),
                               mlp.ref(), _dev(out, torch.float32, (m, H), "out"), m, _stream())
    _lib.check(rc, "mgn_encode")


def decode(x, mlp: MlpWeights, time_dim: int, out_dim: int, delta_norm=None, out_mean_std=None, world_pos=None,
           phi=None, node_type=None, pred=None, next_state=None):
    """``next_state`` = (world_pos [N,2], phi [N,1]) buffers: the last predicted step is also written there
    (``mgn_decode_step``: the state feedback of a rollout step; the buffers may be the current state itself)."""
    n = x.shape[0]
    n_out = time_dim * out_dim
    if mlp.in_dim != H or mlp.out_dim != n_out or n_out > MAX_DEC_OUT:
        raise RuntimeError(f"decode: decoder is {mlp.in_dim} -> {mlp.out_dim}, expected {H} -> {n_out} <= {MAX_DEC_OUT}")
    if next_state is not None:
        if pred is None or delta_norm is not None or out_dim != 3:
            raise RuntimeError("decode: the state feedback goes with pred (out_dim 3) and without delta_norm")
        rc = _lib.lib().mgn_decode_step(
            _dev(x, torch.float32, (n, H), "x"), mlp.ref(), time_dim, _dev(out_mean_std, torch.float32, (2 * n_out,), "out_mean_std"),
            _dev(world_pos, torch.float32, (n, 2), "world_pos"), _dev(phi, torch.float32, (n, 1), "phi"),
            _dev(node_type, torch.float32, (n, 5), "node_type"), _dev(pred, torch.float32, (time_dim, n, 3), "pred"),
            _dev(next_state[0], torch.float32, (n, 2), "next_world"), _dev(next_state[1], torch.float32, (n, 1), "next_phi"),
            n, _stream())
        _lib.check(rc, "mgn_decode_step")
        return
    rc = _lib.lib().mgn_decode(
        _dev(x, torch.float32, (n, H), "x"), mlp.ref(), time_dim, out_dim,
        _opt(delta_norm, torch.float32, (time_dim, n, out_dim), "delta_norm"),
        _opt(out_mean_std, torch.float32, (2 * n_out,), "out_mean_std"), _opt(world_pos, torch.float32, (n, 2), "world_pos"),
        _opt(phi, torch.float32, (n, 1), "phi"), _opt(node_type, torch.float32, (n, 5), "node_type"),
        _opt(pred, torch.float32, (time_dim, n, 3), "pred"), n, _stream())
    _lib.check(rc, "mgn_decode")


def decode_cat(x, x2, w0, b0, w1, b1, time_dim: int, out_dim: int, hidden_ws, delta_norm=None, out_mean_std=None,
               world_pos=None, phi=None, node_type=None, pred=None):
    """:func:`decode` for a decoder whose first Linear reads cat[x, x2] (w0 is [H, 2H])."""
    n = x.shape[0]
    n_out = time_dim * out_dim
    if n_out > MAX_DEC_OUT:
        raise RuntimeError(f"decode_cat: {n_out} outputs exceed {MAX_DEC_OUT}")
    c = MgnMlp(_dev(w0, torch.float32, (H, 2 * H), "w0"), _dev(b0, torch.float32, (H,), "b0"),
               _dev(w1, torch.float32, (n_out, H), "w1"), _dev(b1, torch.float32, (n_out,), "b1"), 0, 0)
    rc = _lib.lib().mgn_decode_cat(
        _dev(x, torch.float32, (n, H), "x"), _dev(x2, torch.float32, (n, H), "x2"), C.byref(c), time_dim, out_dim,
        _dev(hidden_ws, torch.float32, (n, H), "hidden_ws"), _opt(delta_norm, torch.float32, (time_dim, n, out_dim), "delta_norm"),
        _opt(out_mean_std, torch.float32, (2 * n_out,), "out_mean_std"), _opt(world_pos, torch.float32, (n, 2), "world_pos"),
        _opt(phi, torch.float32, (n, 1), "phi"), _opt(node_type, torch.float32, (n, 5), "node_type"),
        _opt(pred, torch.float32, (time_dim, n, 3), "pred"), n, _stream())
    _lib.check(rc, "mgn_decode_cat")


def gather_rows(src, idx, dst):
    m, wd = dst.shape
    rc = _lib.lib().mgn_gather_rows(_dev(src, torch.float32, (None, wd), "src"), _dev(idx, torch.int64, (m,), "idx"),
                                    _dev(dst, torch.float32, (m, wd), "dst"), m, wd, _stream())
    _lib.check(rc, "mgn_gather_rows")


def scatter_rows(src, idx, dst):
    m, wd = src.shape
    rc = _lib.lib().mgn_scatter_rows(_dev(src, torch.float32, (m, wd), "src"), _dev(idx, torch.int64, (m,), "idx"),
                                     _dev(dst, torch.float32, (None, wd), "dst"), m, wd, _stream())
    _lib.check(rc, "mgn_scatter_rows")


# --------------------------------------------------------------------------- device: halo exchange (partitioned meshes)
def halo_pack(src, send_idx, send_buf):
    """send_buf[i] = src[send_idx[i]] (rows of any width; ``send_idx`` int32, every entry a valid row of ``src``)."""
    m, wd = send_buf.shape
    rc = _lib.lib().mgn_halo_pack(_dev(src, torch.float32, (None, wd), "src"), _dev(send_idx, torch.int32, (m,), "send_idx"),
                                  _dev(send_buf, torch.float32, (m, wd), "send_buf"), m, wd, _stream())
    _lib.check(rc, "mgn_halo_pack")


def halo_unpack_add(recv_buf, add_rowptr, add_rows, add_pos, buf):
    """buf[add_rows[d]] += recv_buf[add_pos[k]] for k in add_rowptr[d] .. add_rowptr[d+1]-1, in that order."""
    wd, nd = buf.shape[1], add_rows.shape[0]
    rc = _lib.lib().mgn_halo_unpack_add(
        _dev(recv_buf, torch.float32, (add_pos.shape[0], wd), "recv_buf"), _dev(add_rowptr, torch.int32, (nd + 1,), "add_rowptr"),
        _dev(add_rows, torch.int32, (nd,), "add_rows"), _dev(add_pos, torch.int32, (None,), "add_pos"),
        _dev(buf, torch.float32, (None, wd), "buf"), nd, wd, _stream())
    _lib.check(rc, "mgn_halo_unpack_add")


def halo_stage_state(world_pos, phi, n_owned, send_idx, world_local, phi_local, send_buf):
    """Copies the owned rows of (world_pos, phi) into the local [owned | halo] arrays and writes the
    3-float records (x, y, phi) of the rows listed in ``send_idx`` to ``send_buf``."""
    m = send_idx.shape[0]
    n_loc = world_local.shape[0]
    if world_pos.shape[0] < n_owned or phi.shape[0] < n_owned or n_loc < n_owned:
        raise RuntimeError("halo_stage_state: fewer rows than owned nodes")
    rc = _lib.lib().mgn_halo_stage_state(
        _dev(world_pos, torch.float32, (None, 2), "world_pos"), _dev(phi, torch.float32, (None, 1), "phi"), n_owned,
        _dev(send_idx, torch.int32, (m,), "send_idx"), m, _dev(world_local, torch.float32, (n_loc, 2), "world_local"),
        _dev(phi_local, torch.float32, (n_loc, 1), "phi_local"), _dev(send_buf, torch.float32, (m, 3), "send_buf"), _stream())
    _lib.check(rc, "mgn_halo_stage_state")


def halo_unpack_state(recv_buf, n_owned, world_local, phi_local):
    """Writes the received 3-float records into the halo rows (from row ``n_owned``) of the local arrays."""
    m, n_loc = recv_buf.shape[0], world_local.shape[0]
    if n_owned + m != n_loc:
        raise RuntimeError("halo_unpack_state: owned + received rows must fill the local arrays")
    _dev(world_local, torch.float32, (n_loc, 2), "world_local")
    _dev(phi_local, torch.float32, (n_loc, 1), "phi_local")
    rc = _lib.lib().mgn_halo_unpack_state(_dev(recv_buf, torch.float32, (m, 3), "recv_buf"), m,
                                          C.c_void_p(world_local.data_ptr() + 8 * n_owned),
                                          C.c_void_p(phi_local.data_ptr() + 4 * n_owned), _stream())
    _lib.check(rc, "mgn_halo_unpack_state")


# --------------------------------------------------------------------------- device: coarse graph (multiscale option)
FPS_REGISTER_MAX = 16384          # blocks up to this size need no distance workspace
FPS_ROUNDS_MIN = 30000            # a single mesh beyond this size is sampled with two launches per sample over the whole GPU


def _block_ptr(sizes, device) -> torch.Tensor:
    ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(np.asarray(sizes, dtype=np.int64), out=ptr[1:])
    if ptr[-1] >= 2 ** 31:
        raise RuntimeError("coarse graph: more than 2^31 rows in one batch")
    return torch.from_numpy(ptr.astype(np.int32)).to(device)


def fps(points, block_sizes, sample_counts, starts) -> torch.Tensor:
    """Farthest-point sampling of every block of a block-diagonal batch (``mgn_fps``): ``points`` [N, 2 or 3] float32 on
    the GPU, block b = the next ``block_sizes[b]`` rows, sampled ``sample_counts[b]`` times starting from its row
    ``starts[b]``.  Returns the sampled rows (global ids, int64), block after block."""
    n, dim = points.shape
    sizes, counts = [int(s) for s in block_sizes], [int(c) for c in sample_counts]
    if sum(sizes) != n or len(counts) != len(sizes) or len(starts) != len(sizes):
        raise RuntimeError("fps: block sizes must add up to the number of points, one sample count and start per block")
    for s, c, st in zip(sizes, counts, starts):
        if c < 0 or (c > 0 and not 0 <= int(st) < s):
            raise RuntimeError(f"fps: first sample {int(st)} outside a block of {s} points")
    dev = points.device
    out = torch.empty(sum(counts), dtype=torch.int64, device=dev)
    if not sizes or out.numel() == 0:
        return out
    if len(sizes) == 1 and n >= FPS_ROUNDS_MIN and not torch.cuda.is_current_stream_capturing():
        ws = torch.empty(int(_lib.lib().mgn_fps_workspace_floats(n)), dtype=torch.float32, device=dev)
        rc = _lib.lib().mgn_fps_single(_dev(points, torch.float32, (n, dim), "points"), dim, n, counts[0], int(starts[0]),
                                       ws.data_ptr(), out.data_ptr(), _stream())
        _lib.check(rc, "mgn_fps_single")
        return out
    biggest = max(sizes)
    ws = torch.empty(n, dtype=torch.float32, device=dev) if biggest > FPS_REGISTER_MAX else None
    node_ptr, samp_ptr = _block_ptr(sizes, dev), _block_ptr(counts, dev)
    first = torch.tensor([int(s) for s in starts], dtype=torch.int32).to(dev)
    rc = _lib.lib().mgn_fps(_dev(points, torch.float32, (n, dim), "points"), dim, node_ptr.data_ptr(), samp_ptr.data_ptr(),
                            first.data_ptr(), len(sizes), biggest, _opt(ws, torch.float32, (n,), "dist_ws"), out.data_ptr(),
                            _stream())
    _lib.check(rc, "mgn_fps")
    return out


def knn_graph(points, first, coarse_sizes, k: int) -> torch.Tensor:
    """k-nearest-neighbour graph inside every block of sampled points (``mgn_knn_graph``): the coarse points of block b
    are the rows ``first[...]`` of ``points`` (``first`` None: the rows themselves), ``coarse_sizes[b]`` of them.
    Returns ``edge_index`` [2, E] int64 in coarse numbering: row 0 = neighbour (sender), row 1 = point (receiver)."""
    n, dim = points.shape
    sizes = [int(s) for s in coarse_sizes]
    m = sum(sizes)
    if (first is None and m != n) or (first is not None and first.shape[0] != m):
        raise RuntimeError("knn_graph: block sizes must add up to the number of coarse points")
    dev = points.device
    per_point = [min(int(k), max(s - 1, 0)) for s in sizes]
    eptr = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(np.asarray([s * kk for s, kk in zip(sizes, per_point)], dtype=np.int64), out=eptr[1:])
    out = torch.empty(2, int(eptr[-1]), dtype=torch.int64, device=dev)
    if out.numel() == 0:
        return out
    cptr, eptr_dev = _block_ptr(sizes, dev), torch.from_numpy(eptr).to(dev)
    rc = _lib.lib().mgn_knn_graph(_dev(points, torch.float32, (n, dim), "points"), dim, _opt(first, torch.int64, (m,), "first"),
                                  cptr.data_ptr(), eptr_dev.data_ptr(), len(sizes), max(sizes), int(k), out[0].data_ptr(),
                                  out[1].data_ptr(), _stream())
    _lib.check(rc, "mgn_knn_graph")
    return out


# --------------------------------------------------------------------------- device: training blocks
_MODES = ("f32", "bf16x3", "f16x2")


def matmul_mode() -> str:
    """The matrix path the library is using: "f16x2" (default), "bf16x3" or "f32" (initially from the
    environment variable MGN_MATMUL, read by the library at its first call)."""
    return _MODES[_lib.lib().mgn_get_matmul_mode()]


def set_matmul_mode(mode: str) -> str:
    """Switches the matrix path of all later launches in this process; returns the previous one.  A forward
    pass and its backward pass must run under the same mode."""
    if mode not in _MODES:
        raise ValueError(f"matmul mode must be one of {_MODES}, got {mode!r}")
    prev = matmul_mode()
    _lib.check(_lib.lib().mgn_set_matmul_mode(_MODES.index(mode)), "mgn_set_matmul_mode")
    return prev


def range_check(reset: bool = True) -> bool:
    """True if a kernel on the current device fed the fp16 split an operand of 65520 or more since the last
    reset (the results of that launch chain are then wrong).  Synchronises the current stream."""
    v = C.c_int32(0)
    _lib.check(_lib.lib().mgn_range_check(C.addressof(v), 1 if reset else 0, _stream()), "mgn_range_check")
    return bool(v.value)


def raise_if_out_of_range(what: str):
    """Raises when the fp16-split path saw an operand beyond its range during ``what``."""
    if matmul_mode() == "f16x2" and range_check(reset=True):
        raise RuntimeError(
            f"{what}: an activation or weight reached the range limit of the default matrix path (operands are split "
            "into two fp16 terms: |value| must stay below 65520, weights below 512); the results of this call are "
            "invalid.  Rerun with the environment variable MGN_MATMUL=bf16x3 (or gnn_amd.ops.set_matmul_mode('bf16x3')), "
            "which has no range limit.")


class _RangePoll:
    """Per-device state of the non-blocking range check: a pinned two-word landing buffer and the event that
    marks the copy into it."""
    by_dev = {}

    def __init__(self):
        self.words = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.pending = False


def poll_out_of_range(what: str):
    """Non-blocking form of :func:`raise_if_out_of_range` for loops that must not stall the host (a training loop):
    looks at the answer of the previous poll if the GPU has delivered it -- raising as
    :func:`raise_if_out_of_range` does when the flag was up -- and enqueues the next poll behind the work issued so
    far.  The flag is sticky, so an overflow is reported one call late at worst; finish a loop with
    :func:`raise_if_out_of_range`.  Does nothing while the current stream is being captured."""
    if matmul_mode() != "f16x2" or torch.cuda.is_current_stream_capturing():
        return
    dev = torch.cuda.current_device()
    st = _RangePoll.by_dev.get(dev)
    if st is None:
        st = _RangePoll.by_dev[dev] = _RangePoll()
    if st.pending:
        if not st.event.query():
            return                              # previous answer still on its way: ask again next time
        st.pending = False
        if int(st.words[0]) or int(st.words[1]):
            raise_if_out_of_range(what)         # reads and clears the flag, raises
    _lib.check(_lib.lib().mgn_range_poll(st.words.data_ptr(), _stream()), "mgn_range_poll")
    st.event.record()
    st.pending = True


def train_scratch_bytes() -> int:
    return int(_lib.lib().mgn_train_scratch_bytes())


def _rows128(t, name, rows=None):
    return _dev(t, torch.float32, (rows, H), name)


def _wblock(w, koff, name):
    """[128, ldw] fp32 weight whose columns koff .. koff+127 are used."""
    p = _dev(w, torch.float32, (H, None), name)
    if w.shape[1] % 4 or koff % 4 or koff < 0 or koff + H > w.shape[1]:
        raise RuntimeError(f"{name}: column block {koff}..{koff + H} does not fit a [{H},{w.shape[1]}] weight")
    return p


@_timed_op
def train_linear_fwd(a, w, koff_a, out, b=None, koff_b=0, bias=None, p=None, idx_p=None, q=None, idx_q=None,
                     relu=False):
    rows = a.shape[0]
    if (p is None) != (q is None) or (p is not None and (idx_p is None or idx_q is None)):
        raise RuntimeError("train_linear_fwd: the gathered addends come as (p, idx_p, q, idx_q) or not at all")
    if b is not None and koff_b + H > w.shape[1]:
        raise RuntimeError("train_linear_fwd: second column block out of range")
    rc = _lib.lib().mgn_train_linear_fwd(
        _rows128(a, "A"), _opt(b, torch.float32, (rows, H), "B"), _wblock(w, koff_a, "W"), w.shape[1], koff_a, koff_b,
        _opt(bias, torch.float32, (H,), "bias"), _opt(p, torch.float32, (None, H), "P"),
        _opt(idx_p, torch.int32, (rows,), "idxP"), _opt(q, torch.float32, (None, H), "Q"),
        _opt(idx_q, torch.int32, (rows,), "idxQ"), int(bool(relu)), _rows128(out, "out", rows), rows, _stream())
    _lib.check(rc, "mgn_train_linear_fwd")


@_timed_op
def train_ln_fwd(v, gamma, beta, y=None, res=None, yres=None):
    rows = v.shape[0]
    if (y is None and yres is None) or ((yres is None) != (res is None)):
        raise RuntimeError("train_ln_fwd: give y and/or (res, yres)")
    rc = _lib.lib().mgn_train_ln_fwd(_rows128(v, "v"), _dev(gamma, torch.float32, (H,), "gamma"),
                                     _dev(beta, torch.float32, (H,), "beta"), _opt(res, torch.float32, (rows, H), "res"),
                                     _opt(y, torch.float32, (rows, H), "y"), _opt(yres, torch.float32, (rows, H), "yres"),
                                     rows, _stream())
    _lib.check(rc, "mgn_train_ln_fwd")


class DeferredReductions:
    """Partial-sum space and record list for one backward pass whose weight-gradient reductions run as one launch at
    the end (``mgn_train_defer_reductions`` / ``mgn_train_reduce_many``) instead of one small launch per call -- 88 of
    them in a ten-block step.  Pass the object as the ``scratch`` argument of ``train_ln_bwd`` / ``train_linear_bwd`` /
    ``train_edge_bwd`` inside ``with deferred:``; the gradients those calls name are defined after :meth:`flush`.
    Every call gets partial space of its own from a pool that grows on demand and is kept from step to step (so a
    captured step replays on fixed addresses)."""

    CAPACITY = 512

    def __init__(self, device):
        self.device = torch.device(device)
        self.recs = (_lib.MgnReduceRec * self.CAPACITY)()
        self.count = C.c_int32(0)
        self.chunks, self.chunk, self.offset = [], -1, 0

    def __enter__(self):
        self.count.value, self.chunk, self.offset = 0, (0 if self.chunks else -1), 0
        _lib.check(_lib.lib().mgn_train_defer_reductions(self.recs, self.CAPACITY, C.byref(self.count)), "mgn_train_defer_reductions")
        return self

    def __exit__(self, *exc):
        _lib.lib().mgn_train_defer_reductions(None, 0, None)
        return False

    def take(self, kind: int, rows: int) -> int:
        """Device address of the partial space for the next call of ``kind`` (0 linear_bwd, 1 ln_bwd, 2 edge_bwd)."""
        need = int(_lib.lib().mgn_train_partial_bytes(kind, rows))
        if need < 0:
            raise RuntimeError("mgn_train_partial_bytes: bad arguments")
        need = (need + 255) & ~255
        while self.chunk < 0 or self.offset + need > self.chunks[self.chunk].numel():
            self.chunk, self.offset = self.chunk + 1, 0
            if self.chunk == len(self.chunks):       # a new chunk: at least twice the last one, so large steps need few
                size = max(need, 8 << 20, 2 * self.chunks[-1].numel() if self.chunks else 0)
                self.chunks.append(torch.empty(size, dtype=torch.uint8, device=self.device))
        ptr = self.chunks[self.chunk].data_ptr() + self.offset
        self.offset += need
        return ptr

    def flush(self):
        """Runs all recorded reductions (one launch per 48 records) on the current stream."""
        n, self.count.value = self.count.value, 0
        _lib.check(_lib.lib().mgn_train_reduce_many(self.recs, n, _stream()), "mgn_train_reduce_many")


def _scratch(scratch, kind=0, rows=0):
    if isinstance(scratch, DeferredReductions):
        return scratch.take(kind, rows)
    p = _dev(scratch, torch.uint8, None, "scratch")
    if scratch.numel() < train_scratch_bytes():
        raise RuntimeError("training scratch buffer is smaller than mgn_train_scratch_bytes()")
    return p


@_timed_op
def train_ln_bwd(v, gamma, gpre, dgamma, dbeta, scratch, g=None, gg=None, idx=None, relu_mask=False):
    rows = v.shape[0]
    if g is None and gg is None:
        raise RuntimeError("train_ln_bwd: no upstream gradient")
    if gg is not None and idx is None:
        raise RuntimeError("train_ln_bwd: a gathered gradient needs its row index")
    rc = _lib.lib().mgn_train_ln_bwd(_opt(g, torch.float32, (rows, H), "G"), _opt(gg, torch.float32, (None, H), "Gg"),
                                     _opt(idx, torch.int32, (rows,), "idx"), _rows128(v, "v"),
                                     _dev(gamma, torch.float32, (H,), "gamma"), int(bool(relu_mask)),
                                     _rows128(gpre, "gpre", rows), _dev(dgamma, torch.float32, (H,), "dgamma"),
                                     _dev(dbeta, torch.float32, (H,), "dbeta"), _scratch(scratch, 1, rows), rows, _stream())
    _lib.check(rc, "mgn_train_ln_bwd")


@_timed_op
def train_linear_bwd(g, scratch, a=None, w=None, koff=0, mask=None, res=None, ga=None, dw=None, dw_koff=0, db=None):
    """``dw`` is the full [128, ld] gradient tensor; its columns dw_koff .. dw_koff+127 are written."""
    rows = g.shape[0]
    if ga is None and dw is None:
        raise RuntimeError("train_linear_bwd: nothing to compute")
    if ga is not None and w is None:
        raise RuntimeError("train_linear_bwd: the input gradient needs the weight")
    if dw is not None and a is None:
        raise RuntimeError("train_linear_bwd: the weight gradient needs the layer input")
    dw_ptr, lddw = C.c_void_p(0), 0
    if dw is not None:
        _dev(dw, torch.float32, (H, None), "dW")
        if dw_koff < 0 or dw_koff + H > dw.shape[1]:
            raise RuntimeError("train_linear_bwd: dW column block out of range")
        dw_ptr, lddw = C.c_void_p(dw.data_ptr() + 4 * dw_koff), dw.shape[1]
    rc = _lib.lib().mgn_train_linear_bwd(
        _rows128(g, "G"), _opt(a, torch.float32, (rows, H), "A"),
        C.c_void_p(0) if w is None else _wblock(w, koff, "W"), 0 if w is None else w.shape[1], koff,
        _opt(mask, torch.float32, (rows, H), "mask"), _opt(res, torch.float32, (rows, H), "res"),
        _opt(ga, torch.float32, (rows, H), "GA"), dw_ptr, lddw, _opt(db, torch.float32, (H,), "db"),
        0 if dw is None and isinstance(scratch, DeferredReductions) else _scratch(scratch, 0, rows), rows, _stream())
    _lib.check(rc, "mgn_train_linear_bwd")


@_timed_op
def train_edge_bwd(e_in, g_e, g_aggr, ps, pr, sender, recv, rowptr, w0, w1, b1, ln_w, g1, dpr, dw1, db1, dgamma, dbeta,
                   scratch):
    """Fused backward of one block's edge MLP with recomputed hidden activations (default matrix path)."""
    e, n = e_in.shape[0], dpr.shape[0]
    rc = _lib.lib().mgn_train_edge_bwd(
        _rows128(e_in, "e_in"), _opt(g_e, torch.float32, (e, H), "g_e"), _dev(g_aggr, torch.float32, (None, H), "g_aggr"),
        _dev(ps, torch.float32, (None, H), "ps"), _dev(pr, torch.float32, (None, H), "pr"),
        _dev(sender, torch.int32, (e,), "sender"), _dev(recv, torch.int32, (e,), "recv"),
        _dev(rowptr, torch.int32, (n + 1,), "rowptr"), _dev(w0, torch.float32, (H, 3 * H), "w0"),
        _dev(w1, torch.float32, (H, H), "w1"), _dev(b1, torch.float32, (H,), "b1"), _dev(ln_w, torch.float32, (H,), "ln_w"),
        _rows128(g1, "g1", e), _rows128(dpr, "dpr"), _dev(dw1, torch.float32, (H, H), "dW1"),
        _dev(db1, torch.float32, (H,), "db1"), _dev(dgamma, torch.float32, (H,), "dgamma"),
        _dev(dbeta, torch.float32, (H,), "dbeta"), _scratch(scratch, 2, e), n, e, _stream())
    _lib.check(rc, "mgn_train_edge_bwd")


@_timed_op
def train_attn_edge_bwd(g_aggr, recv, e_out, e_in, edge_weight, g_e_in, g_weight, g_e_out):
    """Attention option: gradient of the message weights [E] and of the edge MLP's (pre-residual) output [E,H]."""
    e, n = e_out.shape[0], g_aggr.shape[0]
    rc = _lib.lib().mgn_train_attn_edge_bwd(
        _dev(g_aggr, torch.float32, (n, H), "g_aggr"), _dev(recv, torch.int32, (e,), "recv"), _rows128(e_out, "e_out", e),
        _opt(e_in, torch.float32, (e, H), "e_in"), _dev(edge_weight, torch.float32, (e,), "edge_weight"), _opt(g_e_in, torch.float32, (e, H), "g_e_in"),
        _dev(g_weight, torch.float32, (e,), "g_weight"), _rows128(g_e_out, "g_e_out", e), e, _stream())
    _lib.check(rc, "mgn_train_attn_edge_bwd")


@_timed_op
def train_attn_alpha_bwd(rowptr, alpha, g_weight, g_dot):
    n, e = rowptr.shape[0] - 1, alpha.shape[0]
    rc = _lib.lib().mgn_train_attn_alpha_bwd(_dev(rowptr, torch.int32, (n + 1,), "rowptr"), _dev(alpha, torch.float32, (e,), "alpha"),
                                             _dev(g_weight, torch.float32, (e,), "g_weight"), _dev(g_dot, torch.float32, (e,), "g_dot"),
                                             n, _stream())
    _lib.check(rc, "mgn_train_attn_alpha_bwd")


@_timed_op
def train_attn_node_grad(a_proj, rowptr, order, other, g_dot, out, accumulate):
    """out[v] (+)= sum over v's edges (``rowptr``, through ``order`` if given) of g_dot[j] * a_proj[other[j]]."""
    n, e = out.shape[0], g_dot.shape[0]
    rc = _lib.lib().mgn_train_attn_node_grad(
        _dev(a_proj, torch.float32, (None, H), "a_proj"), _dev(rowptr, torch.int32, (n + 1,), "rowptr"),
        _opt(order, torch.int32, (e,), "order"), _dev(other, torch.int32, (e,), "other"), _dev(g_dot, torch.float32, (e,), "g_dot"),
        _rows128(out, "out"), 1 if accumulate else 0, n, _stream())
    _lib.check(rc, "mgn_train_attn_node_grad")


@_timed_op
def segment_sum(src, rowptr, out, order=None):
    n = out.shape[0]
    _dev(rowptr, torch.int32, (n + 1,), "rowptr")
    rc = _lib.lib().mgn_segment_sum(_dev(src, torch.float32, (None, H), "src"), C.c_void_p(rowptr.data_ptr()),
                                    _opt(order, torch.int32, (None,), "order"), _rows128(out, "out"), n, _stream())
    _lib.check(rc, "mgn_segment_sum")


@_timed_op
def segment_sum_weighted(src, rowptr, weight, out):
    n, e = out.shape[0], src.shape[0]
    rc = _lib.lib().mgn_segment_sum_weighted(_rows128(src, "src"), _dev(rowptr, torch.int32, (n + 1,), "rowptr"),
                                             _dev(weight, torch.float32, (e,), "weight"), _rows128(out, "out"), n, _stream())
    _lib.check(rc, "mgn_segment_sum_weighted")


@_timed_op
def train_pad_features(feat, mean_std, out):
    rows, f = feat.shape
    if f > H:
        raise RuntimeError("train_pad_features: more than 128 features")
    rc = _lib.lib().mgn_train_pad_features(_dev(feat, torch.float32, (rows, f), "feat"), f,
                                           _opt(mean_std, torch.float32, (2 * f,), "mean_std"),
                                           _rows128(out, "out", rows), rows, _stream())
    _lib.check(rc, "mgn_train_pad_features")


@_timed_op
def train_loss_fwd(decoded, target_norm, node_type, time_dim, loss3, scratch):
    n = decoded.shape[0]
    rc = _lib.lib().mgn_train_loss_fwd(_rows128(decoded, "decoded"),
                                       _dev(target_norm, torch.float32, (time_dim, n, 3), "target_norm"),
                                       _dev(node_type, torch.float32, (n, 5), "node_type"), time_dim, n,
                                       _dev(loss3, torch.float32, (3,), "loss3"), _scratch(scratch), _stream())
    _lib.check(rc, "mgn_train_loss_fwd")


@_timed_op
def train_loss_bwd(decoded, target_norm, node_type, grad3, time_dim, gdecoded):
    n = decoded.shape[0]
    rc = _lib.lib().mgn_train_loss_bwd(_rows128(decoded, "decoded"),
                                       _dev(target_norm, torch.float32, (time_dim, n, 3), "target_norm"),
                                       _dev(node_type, torch.float32, (n, 5), "node_type"),
                                       _dev(grad3, torch.float32, (3,), "grad3"), time_dim, n,
                                       _rows128(gdecoded, "gdecoded", n), _stream())
    _lib.check(rc, "mgn_train_loss_bwd")
