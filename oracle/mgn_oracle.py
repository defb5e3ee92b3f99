"""CPU restatement of the reference MeshGraphNets hot path (test oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Everything here runs on
the host with plain torch / numpy ops and is written functionally over a
``state_dict`` (no ``nn.Module`` classes), so it shares no structure with the
product code it checks.

Reference files restated (paths relative to the reference checkout):
  core/datasetclass.py:7-20      cells_to_edge_index (+ PyG 2.6.1 to_undirected/coalesce)
  core/normalization.py:19-48    Normalizer.forward / inverse / _accumulate / _mean / _std
  core/meshgraphnet.py:33-45     MLP factory
  core/meshgraphnet.py:86-120    EdgeNodeMessagePassing.forward (attention off)
  core/meshgraphnet.py:168-235   forward, _build_node_features, _build_edge_features
  core/meshgraphnet.py:285-317   loss, predict
  core/rollout.py:5-59           rollout

Third-party arithmetic absent from the reference tree and restated from its
published behaviour: torch-scatter 2.1.2 ``scatter_add`` (zeros + scatter_add_
along dim 0) and torch-geometric 2.6.1 ``to_undirected`` (concatenate the
reversed edges, sort by ``row * N + col``, drop duplicates).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "OracleConfig", "NormStats", "cells_to_edge_index", "receiver_csr", "grid_mesh",
    "init_state_dict", "mlp", "mp_layer", "mp_layer_csr_projected", "node_features",
    "edge_features", "forward", "predict", "loss", "rollout", "make_norms",
    "synthetic_graph", "bending_node_type", "state_dict_digest", "forced_relu_masks", "inject_noise",
    "farthest_point_sampling", "knn_graph", "coarse_graph",
]


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Constructor arguments of the reference model (core/meshgraphnet.py:123-134)."""
    node_in_dim: int = 10
    edge_in_dim: int = 7
    hidden_size: int = 128
    process_steps: int = 10
    node_out_dim: int = 3
    with_mat_params: bool = False
    time_dim: int = 1
    attention: bool = False            # core/meshgraphnet.py:60-61, :100-108
    sample_ratio: float = 0.0          # > 0: coarse branch, core/meshgraphnet.py:148-156, :179-194, :237-283
    coarse_process_steps: int = 0


# --------------------------------------------------------------------------- edge build
def cells_to_edge_index(cells: np.ndarray) -> np.ndarray:
    """Triangles -> directed, de-duplicated edge list, sorted by (sender, receiver).

    core/datasetclass.py:12-16 concatenates the three edge blocks (c0->c1, c1->c2,
    c2->c0) column-block-wise; :19 hands them to ``to_undirected`` which appends the
    reversed copy and coalesces on the key ``row * N + col`` with ``N = max id + 1``.
    Integer work: the result must match bit for bit.
    """
    cells = np.asarray(cells, dtype=np.int64).reshape(-1, 3)
    if cells.shape[0] == 0:
        return np.zeros((2, 0), dtype=np.int64)
    a = np.concatenate([cells[:, 0], cells[:, 1], cells[:, 2]])
    b = np.concatenate([cells[:, 1], cells[:, 2], cells[:, 0]])
    row = np.concatenate([a, b])
    col = np.concatenate([b, a])
    n = int(max(row.max(), col.max())) + 1
    key = np.unique(row * n + col)          # sorted ascending, duplicates dropped
    return np.stack([key // n, key % n]).astype(np.int64)


def receiver_csr(edge_index: np.ndarray, num_nodes: int):
    """Receiver-major view of an edge list (not in the reference; the product's layout).

    Returns ``rowptr[N+1]``, ``sender[E]``, ``recv[E]`` (all int32) and ``perm[E]``
    (int64) with ``perm[j]`` = position in ``edge_index`` of the j-th edge of the
    receiver-sorted order.  Ties keep the original relative order (stable sort).
    """
    ei = np.asarray(edge_index, dtype=np.int64)
    snd, rcv = ei[0], ei[1]
    perm = np.argsort(rcv, kind="stable")
    counts = np.bincount(rcv, minlength=num_nodes).astype(np.int64)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return (rowptr.astype(np.int32), snd[perm].astype(np.int32),
            rcv[perm].astype(np.int32), perm.astype(np.int64))


def grid_mesh(nx: int, ny: int, lx: float = 1.0, ly: float = 1.0):
    """Structured triangle mesh: (nx+1)*(ny+1) nodes, ids row-major with x fastest,
    every cell cut along its (v0, v3) diagonal into (v0, v1, v3) and (v0, v2, v3)
    -- the shape dolfin's RectangleMesh produces (gt_gen/gt_gen/bending_gen.py:63-65).
    """
    xs = np.linspace(0.0, lx, nx + 1, dtype=np.float64)
    ys = np.linspace(0.0, ly, ny + 1, dtype=np.float64)
    gx, gy = np.meshgrid(xs, ys, indexing="xy")
    pos = np.stack([gx.ravel(), gy.ravel()], axis=1).astype(np.float32)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (j * (nx + 1) + i).ravel()
    v1, v2, v3 = v0 + 1, v0 + nx + 1, v0 + nx + 2
    cells = np.empty((2 * nx * ny, 3), dtype=np.int64)
    cells[0::2] = np.stack([v0, v1, v3], axis=1)
    cells[1::2] = np.stack([v0, v2, v3], axis=1)
    return pos, cells


# --------------------------------------------------------------------------- weights
def _mlp_keys(prefix: str, idxs):
    return [f"{prefix}.{i}" for i in idxs]


def init_state_dict(cfg: OracleConfig, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Default-initialised weights in the reference's checkpoint layout.

    Modules are created in the order the reference constructor creates them
    (core/meshgraphnet.py:137-158: node encoder, edge encoder, per processor the edge
    MLP then the node MLP, decoder) so that the same ``torch.manual_seed`` yields the
    same tensors.  Keys as listed in selected_model/bending_best/log.txt:2-137.
    """
    H = cfg.hidden_size
    torch.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def lin(key, fin, fout):
        m = torch.nn.Linear(fin, fout)
        sd[key + ".weight"] = m.weight.detach().clone()
        sd[key + ".bias"] = m.bias.detach().clone()

    def ln(key):
        sd[key + ".weight"] = torch.ones(H)
        sd[key + ".bias"] = torch.zeros(H)

    for name, fin in (("node_encoder", cfg.node_in_dim), ("edge_encoder", cfg.edge_in_dim)):
        lin(f"{name}.0", fin, H); lin(f"{name}.2", H, H); ln(f"{name}.3")
    def block(prefix):
        if cfg.attention:                       # created before the two MLPs (core/meshgraphnet.py:60-61)
            lin(f"{prefix}.attn_lin", H, H)
        lin(f"{prefix}.edge_mlp.0", 3 * H, H); lin(f"{prefix}.edge_mlp.2", H, H)
        ln(f"{prefix}.edge_mlp.4")
        lin(f"{prefix}.node_mlp.0", 2 * H, H); lin(f"{prefix}.node_mlp.2", H, H)
        ln(f"{prefix}.node_mlp.4")

    for i in range(cfg.process_steps):
        block(f"processors.{i}")
    if cfg.sample_ratio > 0:                    # core/meshgraphnet.py:148-156
        lin("coarse_edge_encoder.0", cfg.edge_in_dim, H); lin("coarse_edge_encoder.2", H, H); ln("coarse_edge_encoder.3")
        for i in range(cfg.coarse_process_steps):
            block(f"coarse_processors.{i}")
        lin("node_decoder.0", 2 * H, H)
    else:
        lin("node_decoder.0", H, H)
    lin("node_decoder.2", H, cfg.node_out_dim * cfg.time_dim)
    return sd


def state_dict_digest(sd) -> float:
    """Order-dependent float64 checksum used to detect an RNG / init mismatch."""
    tot = 0.0
    for k, (name, t) in enumerate(sd.items()):
        tot += (k + 1) * float(t.double().sum()) + float(t.double().abs().sum())
    return tot


# --------------------------------------------------------------------------- normaliser
@dataclass
class NormStats:
    """Online feature statistics (core/normalization.py:7-17)."""
    acc_sum: torch.Tensor
    acc_sum_squared: torch.Tensor
    acc_count: int = 0
    num_accumulations: int = 0
    max_accumulations: int = 10 ** 6
    std_epsilon: float = 1e-8

    @staticmethod
    def fresh(batch: int, feat: int) -> "NormStats":
        return NormStats(torch.zeros(batch, feat), torch.zeros(batch, feat))

    def clone(self) -> "NormStats":
        return NormStats(self.acc_sum.clone(), self.acc_sum_squared.clone(), self.acc_count,
                         self.num_accumulations, self.max_accumulations, self.std_epsilon)

    def mean(self) -> torch.Tensor:                      # core/normalization.py:39-41
        return self.acc_sum / max(self.acc_count, 1)

    def std(self) -> torch.Tensor:                       # core/normalization.py:43-48
        var = self.acc_sum_squared / max(self.acc_count, 1) - self.mean() ** 2
        return torch.sqrt(torch.clamp(var, min=0.0) + self.std_epsilon)

    def normalize(self, x: torch.Tensor, accumulate: bool = True) -> torch.Tensor:
        """core/normalization.py:19-24 with :30-37 inlined; ``x`` is [B, n, F]."""
        if accumulate and self.num_accumulations < self.max_accumulations:
            self.acc_sum = self.acc_sum + torch.sum(x, dim=1)
            self.acc_sum_squared = self.acc_sum_squared + torch.sum(x ** 2, dim=1)
            self.acc_count += x.shape[1]
            self.num_accumulations += 1
        return (x - self.mean().unsqueeze(1)) / self.std().unsqueeze(1)

    def inverse(self, y: torch.Tensor) -> torch.Tensor:  # core/normalization.py:26-28
        return torch.einsum("ij,ikj->ikj", self.std(), y) + self.mean().unsqueeze(1)


def make_norms(cfg: OracleConfig) -> Dict[str, NormStats]:
    """Three fresh normalisers as the constructor makes them (core/meshgraphnet.py:165-167)."""
    norms = {"node": NormStats.fresh(1, cfg.node_in_dim),
             "edge": NormStats.fresh(1, cfg.edge_in_dim),
             "output": NormStats.fresh(cfg.time_dim, cfg.node_out_dim)}
    if cfg.sample_ratio > 0:                      # core/meshgraphnet.py:150
        norms["coarse_edge"] = NormStats.fresh(1, cfg.edge_in_dim)
    return norms


# --------------------------------------------------------------------------- network
_FORCED_MASKS = None


class forced_relu_masks:
    """Test hook: inside this context every ReLU of the network, in call order (node encoder,
    edge encoder, per block edge MLP x2 then node MLP x2, decoder), multiplies by the next given
    0/1 mask instead of deciding from its input.  With the activation pattern pinned the
    network is smooth, so gradients computed elsewhere under the same pattern can be compared
    tightly.  Outside the context the oracle is unchanged."""

    def __init__(self, masks):
        self.masks = list(masks)

    def __enter__(self):
        global _FORCED_MASKS
        _FORCED_MASKS = iter(self.masks)
        return self

    def __exit__(self, *exc):
        global _FORCED_MASKS
        _FORCED_MASKS = None
        return False


def _relu(h: torch.Tensor) -> torch.Tensor:
    if _FORCED_MASKS is None:
        return F.relu(h)
    return h * next(_FORCED_MASKS).to(h.dtype)


def mlp(sd, prefix: str, x: torch.Tensor, *, final_relu: bool, layer_norm: bool) -> torch.Tensor:
    """Two-Linear MLP as both factories build it (core/meshgraphnet.py:33-45, :64-78):
    Linear(.0) ReLU Linear(.2) [ReLU] [LayerNorm at the next free index]."""
    h = _relu(F.linear(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"]))
    h = F.linear(h, sd[f"{prefix}.2.weight"], sd[f"{prefix}.2.bias"])
    nxt = 3
    if final_relu:
        h = _relu(h); nxt = 4
    if layer_norm:
        h = F.layer_norm(h, (h.shape[-1],), sd[f"{prefix}.{nxt}.weight"], sd[f"{prefix}.{nxt}.bias"], 1e-5)
    return h


def mp_layer(sd, i, x: torch.Tensor, edge_index: torch.Tensor, e: torch.Tensor):
    """One processor block (core/meshgraphnet.py:92-120).  ``i`` is the index into ``processors`` or a full key
    prefix such as ``"coarse_processors.1"``.

    The aggregated message is the pre-residual edge output (:110-113); the node MLP
    sees the old node latents (:116); residuals are added last (:118, :120).  With an
    ``attn_lin`` in the block (:60-61) the messages are weighted as :100-108 does: the leaky-ReLU of the dot
    product of the two projected end points, divided by its sum over the receiver's in-edges (clamped at 1e-6).
    """
    p = i if isinstance(i, str) else f"processors.{i}"
    snd, rcv = edge_index[0], edge_index[1]
    e_new = mlp(sd, f"{p}.edge_mlp", torch.cat([x[snd], x[rcv], e], dim=-1),
                final_relu=True, layer_norm=True)
    msg = e_new
    if f"{p}.attn_lin.weight" in sd:
        a = F.linear(x, sd[f"{p}.attn_lin.weight"], sd[f"{p}.attn_lin.bias"])
        alpha = F.leaky_relu((a[snd] * a[rcv]).sum(dim=-1))
        denom = alpha.new_zeros(x.shape[0]).scatter_add_(0, rcv, alpha)
        denom[rcv] = denom[rcv].clamp(min=1e-6)
        msg = e_new * (alpha / denom[rcv]).unsqueeze(-1)
    aggr = e_new.new_zeros(x.shape[0], e_new.shape[1])
    aggr.scatter_add_(0, rcv.view(-1, 1).expand_as(msg), msg)
    x_new = mlp(sd, f"{p}.node_mlp", torch.cat([x, aggr], dim=-1),
                final_relu=True, layer_norm=True)
    return x_new + x, e_new + e


def farthest_point_sampling(points: torch.Tensor, num_samples: int, start: int) -> torch.Tensor:
    """core/meshgraphnet.py:243-254 with the random first index (:248) passed in."""
    n = points.shape[0]
    idx = torch.zeros(num_samples, dtype=torch.long)
    idx[0] = start
    dist = torch.full((n,), float("inf"))
    for k in range(1, num_samples):
        last = points[idx[k - 1]].unsqueeze(0)
        dist = torch.minimum(dist, torch.norm(points - last, dim=-1))
        idx[k] = torch.argmax(dist)
    return idx


def knn_graph(points: torch.Tensor, k: int) -> torch.Tensor:
    """Restatement of ``torch_geometric.nn.knn_graph(x, k, loop=False)`` (torch-cluster 1.6.3, absent here; no
    vector of the reference pins it): for every point its k nearest other points by Euclidean distance, nearest
    first, ties by lower index; edge = (neighbour -> point), edges grouped by point in ascending point order.
    Squared distances are formed coordinate by coordinate in float64, so equal distances compare equal."""
    n = points.shape[0]
    p = points.double()
    diff = p.unsqueeze(1) - p.unsqueeze(0)
    d2 = (diff * diff).sum(dim=-1) if p.shape[1] != 2 else diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]
    d2.fill_diagonal_(float("inf"))
    kk = min(k, n - 1)
    order = torch.argsort(d2, dim=1, stable=True)[:, :kk]
    centre = torch.arange(n).unsqueeze(1).expand(n, kk)
    return torch.stack([order.reshape(-1), centre.reshape(-1)])


def coarse_graph(cfg: OracleConfig, g, start: int):
    """(coarse edge features [Ec,7], coarse edge_index [2,Ec] in coarse numbering, first_indices [Nc]):
    core/meshgraphnet.py:237-283."""
    mesh_pos, world_pos, phi = g["mesh_pos"], g["world_pos"], g["phi"]
    k = 4 if mesh_pos.shape[-1] == 3 else 3
    n_coarse = max(1, int(mesh_pos.shape[0] * cfg.sample_ratio))
    # the sample and its graph are integer results of float32 arithmetic in the reference: kept in float32 when the
    # rest of the oracle runs in float64 (gradient checks), so that both precisions see one topology
    first = farthest_point_sampling(mesh_pos.float(), n_coarse, start)
    cei = knn_graph(mesh_pos.float()[first], k)
    s, r = first[cei[0]], first[cei[1]]
    rel = mesh_pos[s] - mesh_pos[r]
    relw = world_pos[s] - world_pos[r]
    feats = torch.cat([rel, torch.norm(rel, dim=-1, keepdim=True), relw, torch.norm(relw, dim=-1, keepdim=True),
                       phi[s] - phi[r]], dim=-1)
    return feats, cei, first


def mp_layer_csr_projected(sd, i: int, x: torch.Tensor, rowptr, sender, recv, e_csr: torch.Tensor):
    """The same block in the product's formulation, in float64, for cross-checking the
    algebra the kernels use: edges in receiver-major order, the first edge Linear split
    into per-node projections ``x @ W1[:, :H].T`` (sender) and ``x @ W1[:, H:2H].T + b1``
    (receiver) plus ``e @ W1[:, 2H:].T``.  Mathematically identical to :func:`mp_layer`.
    """
    H = x.shape[1]
    d = lambda k: sd[k].double()
    p = f"processors.{i}"
    W1 = d(f"{p}.edge_mlp.0.weight")
    xs = x.double()
    ps = xs @ W1[:, :H].T
    pr = xs @ W1[:, H:2 * H].T + d(f"{p}.edge_mlp.0.bias")
    s = torch.as_tensor(np.asarray(sender), dtype=torch.long)
    r = torch.as_tensor(np.asarray(recv), dtype=torch.long)
    h1 = F.relu(ps[s] + pr[r] + e_csr.double() @ W1[:, 2 * H:].T)
    h2 = F.relu(F.linear(h1, d(f"{p}.edge_mlp.2.weight"), d(f"{p}.edge_mlp.2.bias")))
    e_new = F.layer_norm(h2, (H,), d(f"{p}.edge_mlp.4.weight"), d(f"{p}.edge_mlp.4.bias"), 1e-5)
    aggr = torch.zeros_like(xs)
    aggr.index_add_(0, r, e_new)
    g = F.relu(F.linear(torch.cat([xs, aggr], -1), d(f"{p}.node_mlp.0.weight"), d(f"{p}.node_mlp.0.bias")))
    g = F.relu(F.linear(g, d(f"{p}.node_mlp.2.weight"), d(f"{p}.node_mlp.2.bias")))
    x_new = F.layer_norm(g, (H,), d(f"{p}.node_mlp.4.weight"), d(f"{p}.node_mlp.4.bias"), 1e-5) + xs
    return x_new, e_new + e_csr.double()


def node_features(cfg: OracleConfig, g) -> torch.Tensor:
    """[u, phi, swelling_phi, node_type(, mat_param)] (core/meshgraphnet.py:212-225)."""
    u = g["world_pos"] - g["mesh_pos"]
    parts = [u, g["phi"], g["swelling_phi"], g["node_type"]]
    if cfg.with_mat_params:
        parts.append(g["mat_param"].unsqueeze(0).repeat(u.shape[0], 1))
    return torch.cat(parts, dim=-1)


def edge_features(g) -> torch.Tensor:
    """[dX, |dX|, dx, |dx|, dphi] sender minus receiver (core/meshgraphnet.py:226-235)."""
    s, r = g["edge_index"][0], g["edge_index"][1]
    dm = g["mesh_pos"][s, :] - g["mesh_pos"][r, :]
    dw = g["world_pos"][s, :] - g["world_pos"][r, :]
    return torch.cat([dm, torch.norm(dm, dim=-1, keepdim=True),
                      dw, torch.norm(dw, dim=-1, keepdim=True),
                      g["phi"][s] - g["phi"][r]], dim=-1)


def forward(sd, cfg: OracleConfig, norms: Dict[str, NormStats], g, *, return_latents: bool = False, coarse_start: int = 0):
    """Normalised delta prediction [T, N, out] (core/meshgraphnet.py:168-211, fine path
    only: sample_ratio == 0).  Both feature normalisers accumulate on every call."""
    x = node_features(cfg, g).unsqueeze(0)
    e = edge_features(g).unsqueeze(0)
    x_h = mlp(sd, "node_encoder", norms["node"].normalize(x), final_relu=False, layer_norm=True).squeeze(0)
    e_h = mlp(sd, "edge_encoder", norms["edge"].normalize(e), final_relu=False, layer_norm=True).squeeze(0)
    x0, e0 = x_h, e_h
    if cfg.sample_ratio > 0:                                    # core/meshgraphnet.py:179-182
        ce, cei, first = coarse_graph(cfg, g, coarse_start)
        ce_h = mlp(sd, "coarse_edge_encoder", norms["coarse_edge"].normalize(ce.unsqueeze(0)), final_relu=False,
                   layer_norm=True).squeeze(0)
        cx_h = x_h[first]
    for i in range(cfg.process_steps):
        x_h, e_h = mp_layer(sd, i, x_h, g["edge_index"], e_h)
    if cfg.sample_ratio > 0:                                    # :187-194
        for i in range(cfg.coarse_process_steps):
            cx_h, ce_h = mp_layer(sd, f"coarse_processors.{i}", cx_h, cei, ce_h)
        map_back = torch.zeros_like(x_h)
        map_back[first] = cx_h
        x_h = torch.cat([x_h, map_back], dim=-1)
    dec = mlp(sd, "node_decoder", x_h, final_relu=False, layer_norm=False)
    delta = dec.view(-1, cfg.time_dim, cfg.node_out_dim).permute(1, 0, 2)
    dt = torch.arange(1, cfg.time_dim + 1).view(cfg.time_dim, 1, 1)
    out = delta * dt
    if return_latents:
        return out, {"x_enc": x0, "e_enc": e0, "x_proc": x_h, "e_proc": e_h}
    return out


def predict(sd, cfg: OracleConfig, norms: Dict[str, NormStats], g, coarse_start: int = 0) -> torch.Tensor:
    """Next state(s) [T, N, 3] (core/meshgraphnet.py:305-317)."""
    with torch.no_grad():
        delta = norms["output"].inverse(forward(sd, cfg, norms, g, coarse_start=coarse_start))
        cur = torch.cat([g["world_pos"], g["phi"]], dim=-1)
        for k in range(3):
            delta[:, g["node_type"][:, k + 1] == 1, k] = 0.0
        return cur.unsqueeze(0) + delta


def loss(sd, cfg: OracleConfig, norms: Dict[str, NormStats], g, coarse_start: int = 0):
    """Masked squared error on normalised deltas (core/meshgraphnet.py:285-304).
    The output normaliser accumulates before the forward pass runs."""
    cur = torch.cat([g["world_pos"], g["phi"]], dim=-1)
    tgt = norms["output"].normalize(g["target"] - cur)
    err = (forward(sd, cfg, norms, g, coarse_start=coarse_start) - tgt) ** 2
    parts = []
    for k in range(3):
        keep = ~(g["node_type"][:, k + 1] == 1)
        sl = slice(k, k + 1) if k < 2 else slice(2, None)
        parts.append(torch.mean(torch.sum(torch.sum(err[:, keep, sl], dim=2), dim=1)))
    return parts[0] + parts[1] + parts[2], parts[0], parts[1], parts[2]


def rollout(sd, cfg: OracleConfig, norms: Dict[str, NormStats], static, world_pos, phi, swelling, coarse_start: int = 0):
    """Autoregressive rollout (core/rollout.py:5-45).

    ``static`` holds mesh_pos / node_type / mat_param / edge_index; ``world_pos`` [T,N,2],
    ``phi`` [T,N,1] and ``swelling`` [T,N] are the ground-truth frames.  Frame 0 seeds
    the state, every step sees the swelling window ``t .. t+time_dim`` (zero past the
    end, :17-26) and feeds its last prediction back (:29-32).
    Returns (pred [T,N,3], rmse[3]).
    """
    T, Td = world_pos.shape[0], cfg.time_dim
    g = dict(static)
    g["world_pos"], g["phi"] = world_pos[0].clone(), phi[0].clone()
    out = [torch.cat([world_pos[0], phi[0]], dim=-1).unsqueeze(0)]
    for t in range(0, T, Td):
        cols = [swelling[t + w] if t + w < T else torch.zeros_like(swelling[0]) for w in range(Td + 1)]
        g["swelling_phi"] = torch.stack(cols).T
        p = predict(sd, cfg, norms, g, coarse_start=coarse_start)
        last = p[-1].clone()
        g["world_pos"], g["phi"] = last[:, :2], last[:, 2:]
        out.append(p)
    pred = torch.cat(out, dim=0)[:T]
    gt = torch.cat([world_pos, phi], dim=-1)
    err = (pred - gt) ** 2
    rmse = torch.stack([torch.sqrt(torch.mean(err[:, :, k])) for k in range(3)])
    return pred, rmse


# --------------------------------------------------------------------------- synthetic inputs
def inject_noise(world_pos, phi, node_type, z_world, z_phi, level: float, conn_length, phi_range):
    """Training noise of one frame with the standard-normal draws given (core/datasetclass.py:75-89)."""
    wn = z_world * level * conn_length
    wn[node_type[:, 1] == 1, 0] = 0.0
    wn[node_type[:, 2] == 1, 1] = 0.0
    pn = z_phi * level * phi_range
    pn[node_type[:, 3] == 1] = 0.0
    return world_pos + wn, phi + pn


def bending_node_type(nx: int, ny: int) -> np.ndarray:
    """Five indicator columns in the layout of gt_gen/gt_gen/bending_gen.py:11-16, :317-322:
    column 0 free, 1 ux fixed, 2 uy fixed, 3 phi fixed, 4 swelling boundary.  Left column of
    nodes: ux and uy fixed; top row: swelling; everything else free."""
    n = (nx + 1) * (ny + 1)
    ids = np.arange(n)
    left = (ids % (nx + 1)) == 0
    top = (ids // (nx + 1)) == ny
    nt = np.zeros((n, 5), dtype=np.float32)
    nt[left, 1] = 1.0; nt[left, 2] = 1.0
    nt[top, 4] = 1.0
    nt[~(left | top), 0] = 1.0
    return nt


def synthetic_graph(nx: int, ny: int, seed: int = 0, lx: float = 1.0, ly: float = 1.0,
                    time_dim: int = 1, with_target: bool = False):
    """Seeded synthetic state on a structured triangle mesh: world_pos = mesh_pos + 1e-3 N(0,1),
    phi = 0.5 + 0.1 U(0,1), swelling 0.35 on the top row (all window columns), bending-style
    node types, mat_param = [0.2, 7.5e-6]."""
    pos, cells = grid_mesh(nx, ny, lx, ly)
    n = pos.shape[0]
    gen = torch.Generator().manual_seed(seed)
    mesh_pos = torch.from_numpy(pos)
    nt = torch.from_numpy(bending_node_type(nx, ny))
    g = {
        "mesh_pos": mesh_pos,
        "world_pos": mesh_pos + 1e-3 * torch.randn(n, 2, generator=gen),
        "phi": 0.5 + 0.1 * torch.rand(n, 1, generator=gen),
        "swelling_phi": 0.35 * nt[:, 4:5].repeat(1, time_dim + 1),
        "node_type": nt,
        "mat_param": torch.tensor([0.2, 7.5e-6]),
        "edge_index": torch.from_numpy(cells_to_edge_index(cells)),
        "cells": torch.from_numpy(cells),
    }
    if with_target:
        cur = torch.cat([g["world_pos"], g["phi"]], dim=-1)
        g["target"] = cur.unsqueeze(0) + 1e-3 * torch.randn(time_dim, n, 3, generator=gen)
    return g
