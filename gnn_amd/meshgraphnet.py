"""Encode-process-decode MeshGraphNets model on the MI355X HIP kernels.

Public surface and checkpoint layout follow ``core/meshgraphnet.py`` of the reference:
``MLP`` (:33-45), ``EdgeNodeMessagePassing`` (:46-120) and
``EncodeProcessDecodeMultiScale`` (:122-332) with ``forward`` / ``loss`` / ``predict`` /
``save_model`` / ``load_model``.  Parameters sit in the same ``nn.Sequential`` slots, so
``state_dict`` has the reference's 136 keys and a seeded construction draws the same
weights.  All arithmetic runs in libmgn_hip.so; there is no PyTorch or CPU fallback.  The attention option
(:60-61, :100-108) and the coarse multiscale branch (:148-156, :179-194, :237-283), off in every shipped config,
run in ``forward`` / ``predict`` / ``rollout`` and in ``loss`` + backward.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import H
from .graph import Topology, TopologyCache
from .normalization import Normalizer, ensure_core_alias, load_normalizer

ensure_core_alias()


class KernelMLP(nn.Sequential):
    """``nn.Sequential`` parameter layout of the reference's MLPs, evaluated by the HIP
    encoder / decoder kernels.  Only the two-Linear shapes the model uses are supported."""

    def _ln_index(self) -> Optional[int]:
        cached = self.__dict__.get("_mgn_ln", -1)
        if cached == -1:
            cached = None
            for i, m in enumerate(self):
                if isinstance(m, nn.LayerNorm):
                    cached = i
            self.__dict__["_mgn_ln"] = cached
        return cached

    def weights(self) -> ops.MlpWeights:
        """Device pointers of the MLP's parameters, rebuilt only when a parameter was replaced or
        moved (this runs twice per block and step: the check reads the six pointers directly)."""
        mods, ln = self._modules, self._ln_index()
        p0, p2 = mods["0"]._parameters, mods["2"]._parameters
        key = (p0["weight"].data_ptr(), p0["bias"].data_ptr(), p2["weight"].data_ptr(), p2["bias"].data_ptr())
        if ln is not None:
            pl = mods[str(ln)]._parameters
            key += (pl["weight"].data_ptr(), pl["bias"].data_ptr())
        cached = self.__dict__.get("_mgn_w")
        if cached is None or cached[0] != key:
            cached = (key, ops.MlpWeights.from_sequential(self, ln))
            self.__dict__["_mgn_w"] = cached
        return cached[1]

    def forward(self, x):
        lead = x.shape[:-1]
        rows = x.reshape(-1, x.shape[-1]).contiguous()
        w = self.weights()
        if self._ln_index() is not None and len(self) == 4:          # encoder: Linear ReLU Linear LN
            out = torch.empty(rows.shape[0], H, dtype=torch.float32, device=rows.device)
            ops.encode(rows, None, w, out)
            return out.reshape(*lead, H)
        if self._ln_index() is None and len(self) == 3:              # decoder: Linear ReLU Linear
            out = torch.empty(1, rows.shape[0], w.out_dim, dtype=torch.float32, device=rows.device)
            ops.decode(rows, w, 1, w.out_dim, delta_norm=out)
            return out.reshape(*lead, w.out_dim)
        raise NotImplementedError("KernelMLP.forward: only the encoder and decoder shapes run stand-alone; "
                                  "the processor MLPs run inside EdgeNodeMessagePassing")

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_mgn_w", None)
        d.pop("_mgn_ln", None)
        return d


def MLP(in_dim, out_dim, hidden_dims=(128, 128), activate_final=False, layer_norm=False):
    """Factory with the reference's signature (core/meshgraphnet.py:33-45)."""
    if len(tuple(hidden_dims)) != 1 or hidden_dims[0] != H:
        raise NotImplementedError(f"MLP: the HIP kernels implement one hidden layer of width {H}, got {hidden_dims}")
    layers = [nn.Linear(in_dim, H), nn.ReLU(), nn.Linear(H, out_dim)]
    if activate_final:
        layers.append(nn.ReLU())
    if layer_norm:
        layers.append(nn.LayerNorm(out_dim))
    return KernelMLP(*layers)


def _on_graph_device(method):
    """Runs a model method with the graph's GPU as the current device: the kernels are enqueued on the current
    device's stream, so a model on cuda:1 called while cuda:0 is current would otherwise launch on the wrong GPU."""
    import functools

    @functools.wraps(method)
    def wrapper(self, graph, *args, **kwargs):
        dev = graph["mesh_pos"].device
        if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                return method(self, graph, *args, **kwargs)
        return method(self, graph, *args, **kwargs)
    return wrapper


class _Workspace:
    """Device buffers for one (device, N, E) problem size, reused across calls."""

    def __init__(self, device, n, e, node_f):
        f32 = dict(dtype=torch.float32, device=device)
        self.key = (str(device), n, e, node_f)
        self.x = torch.empty(n, H, **f32)
        self.e = torch.empty(e, H, **f32)
        self.ps = torch.empty(n, H, **f32)
        self.pr = torch.empty(n, H, **f32)
        self.aggr = torch.empty(n, H, **f32)
        self.node_feat = torch.empty(n, node_f, **f32)
        self.edge_feat = torch.empty(e, 7, **f32)


class EdgeNodeMessagePassing(nn.Module):
    """One processor block (core/meshgraphnet.py:46-120): edge update from
    (sender, receiver, edge) latents, receiver-wise sum of the new edge latents, node
    update, both residuals."""

    def __init__(self, hidden_dim, attention=False):
        super().__init__()
        if hidden_dim != H:
            raise NotImplementedError(f"EdgeNodeMessagePassing: hidden_dim must be {H} for the HIP kernels")
        self.hidden_dim = hidden_dim
        self.attention = attention
        if self.attention:                  # created before the two MLPs, as the reference does (core/meshgraphnet.py:60-61)
            self.attn_lin = nn.Linear(hidden_dim, hidden_dim)
        self.edge_mlp = KernelMLP(nn.Linear(hidden_dim * 3, hidden_dim), nn.ReLU(),
                                  nn.Linear(hidden_dim, hidden_dim), nn.ReLU(), nn.LayerNorm(hidden_dim))
        self.node_mlp = KernelMLP(nn.Linear(hidden_dim * 2, hidden_dim), nn.ReLU(),
                                  nn.Linear(hidden_dim, hidden_dim), nn.ReLU(), nn.LayerNorm(hidden_dim))
        self.__dict__["_topo"] = TopologyCache()

    def step_csr(self, x, e_csr, topo: Topology, ps, pr, aggr, halo=None, projected=False, next_block=None,
                 store_edges=True, packed=None):
        """In place on ``x`` [N,H] and ``e_csr`` [E,H] (receiver-major edge order).

        ``projected``: ``ps``/``pr`` already hold this block's per-node projections (written by
        the previous block's fused node kernel).  ``next_block``: the block that runs next; its
        projections are then produced by this block's node kernel.  ``store_edges=False``: the
        updated edge latents are not written (last block of a stack).  With a ``halo`` plan only
        the first ``halo.num_owned`` rows of ``x`` are owned; the sender projections of the
        remaining (halo) rows arrive from their owners.  ``packed``: this block's (edge, node) entries of an
        ``ops.PackedWeights`` refreshed in this call."""
        ew, nw = self.edge_mlp.weights(), self.node_mlp.weights()
        n = x.shape[0] if halo is None else halo.num_owned
        e_out = e_csr if store_edges else None
        wts = None
        if self.attention:
            # message weights from the OLD node latents (core/meshgraphnet.py:101-107): a = attn_lin(x), then per edge
            # leaky_relu(<a[s], a[r]>) over its sum per receiver.  On a partitioned mesh the projected rows of the halo
            # nodes come from their owners (every receiver is owned, so its sum is complete on this rank).
            buf = self.__dict__.get("_att")
            if buf is None or buf[0].shape[0] != x.shape[0] or buf[1].shape[0] != e_csr.shape[0] or buf[0].device != x.device:
                buf = (torch.empty_like(x), torch.empty(e_csr.shape[0], dtype=torch.float32, device=x.device),
                       torch.empty(e_csr.shape[0], dtype=torch.float32, device=x.device))
                self.__dict__["_att"] = buf
            a_proj, alpha, wts = buf
            ops.train_linear_fwd(x if halo is None else x[:n], self.attn_lin.weight.data, 0,
                                 a_proj if halo is None else a_proj[:n], bias=self.attn_lin.bias.data)
            if halo is not None:
                halo.exchange_rows(a_proj)
            ops.edge_attention(a_proj, topo.rowptr, topo.sender, topo.recv, alpha, wts)
        nxt = next_block.edge_mlp._modules["0"]._parameters if next_block is not None else None
        if halo is None:
            xo, ao, rp, pso, pro = x, aggr, topo.rowptr, ps, pr
        else:                                     # the rows this rank owns
            xo, ao, rp, pso, pro = x[:n], aggr[:n], topo.rowptr[: n + 1], ps[:n], pr[:n]
        if not projected:
            with ops.timed("mp_project"):
                ops.mp_project(xo, self.edge_mlp[0].weight.data, self.edge_mlp[0].bias.data, pso, pro)
        pk_e = packed[0] if packed is not None else None
        if halo is None:
            with ops.timed("mp_edge"):
                ops.mp_edge(e_csr, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, e_out, aggr, edge_weight=wts, packed=pk_e)
        else:
            # the receivers whose in-edges all come from owned senders are updated while the halo
            # rows of `ps` are in flight; the rest follows once they have arrived
            with ops.timed("halo_exchange"):
                pending = halo.exchange_rows(ps, async_op=True)
            with ops.timed("mp_edge"):
                ops.mp_edge(e_csr, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, e_out, aggr, node_range=halo.interior,
                            edge_weight=wts, packed=pk_e)
            if pending is not None:
                pending.wait()
            with ops.timed("mp_edge_boundary"):
                for rng in halo.boundary:
                    ops.mp_edge(e_csr, ps, pr, topo.rowptr, topo.sender, topo.recv, ew, e_out, aggr, node_range=rng,
                                edge_weight=wts, packed=pk_e)
        with ops.timed("mp_node"):
            if packed is not None:
                if nxt is None:
                    ops.mp_node_packed(xo, ao, rp, nw, packed[1], xo)
                else:
                    ops.mp_node_packed(xo, ao, rp, nw, packed[1], xo, nxt["weight"].data, nxt["bias"].data, pso, pro)
            elif nxt is None:
                ops.mp_node(xo, ao, rp, nw, xo)
            else:
                ops.mp_node_project(xo, ao, rp, nw, nxt["weight"].data, nxt["bias"].data, xo, pso, pro)

    def forward(self, node_feat, edge_index, edge_feat):
        """node_feat [N,H], edge_index [2,E] (row 0 sender, row 1 receiver), edge_feat [E,H]
        in ``edge_index`` order; returns the updated (node_feat, edge_feat) in the same order."""
        n, e = node_feat.shape[0], edge_feat.shape[0]
        if "_topo" not in self.__dict__:
            self.__dict__["_topo"] = TopologyCache()
        topo = self.__dict__["_topo"].get(edge_index, n, node_feat.device)
        x = node_feat.detach().to(torch.float32).contiguous().clone()
        e_csr = torch.empty(e, H, dtype=torch.float32, device=x.device)
        ops.gather_rows(edge_feat.detach().to(torch.float32).contiguous(), topo.perm, e_csr)
        ps, pr, aggr = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        self.step_csr(x, e_csr, topo, ps, pr, aggr)
        e_out = torch.empty_like(e_csr)
        ops.scatter_rows(e_csr, topo.perm, e_out)
        return x, e_out


class EncodeProcessDecodeMultiScale(nn.Module):
    def __init__(self,
                 node_in_dim,
                 edge_in_dim,
                 hidden_size=128,
                 process_steps=3,
                 coarse_process_steps=3,
                 node_out_dim=0,
                 attention=False,
                 with_mat_params=True,
                 sample_ratio=0.5,
                 time_dim=1,
                 device="cpu"):
        super().__init__()
        if hidden_size != H:
            raise NotImplementedError(f"EncodeProcessDecodeMultiScale: hidden_size must be {H}")
        if edge_in_dim != 7:
            raise NotImplementedError("EncodeProcessDecodeMultiScale: edge features are the 7 the reference builds")
        self.node_encoder = MLP(node_in_dim, hidden_size, hidden_dims=(hidden_size,), layer_norm=True)
        self.edge_encoder = MLP(edge_in_dim, hidden_size, hidden_dims=(hidden_size,), layer_norm=True)
        self.attention = attention
        self.with_mat_params = with_mat_params
        self.sample_ratio = sample_ratio
        self.time_dim = time_dim
        self.node_out_dim = node_out_dim
        self.process_steps = process_steps
        self.coarse_process_steps = coarse_process_steps
        self.processors = nn.ModuleList([EdgeNodeMessagePassing(hidden_size, attention) for _ in range(process_steps)])
        if sample_ratio > 0:                # coarse branch, modules in the reference's order (core/meshgraphnet.py:148-156)
            self.coarse_edge_encoder = MLP(edge_in_dim, hidden_size, hidden_dims=(hidden_size,), layer_norm=True)
            self.coarse_edge_features_normalizer = Normalizer(1, edge_in_dim, "edge_features_normalizer", device)
            self.coarse_processors = nn.ModuleList([EdgeNodeMessagePassing(hidden_size, attention)
                                                    for _ in range(coarse_process_steps)])
            self.node_decoder = nn.Sequential(nn.Linear(2 * hidden_size, hidden_size), nn.ReLU(),
                                              nn.Linear(hidden_size, node_out_dim * time_dim))
        else:
            self.node_decoder = MLP(hidden_size, node_out_dim * time_dim, hidden_dims=(hidden_size,), layer_norm=False)
        # first sample of the coarse graph's farthest-point sampling: None = drawn anew on every call, as the
        # reference does (core/meshgraphnet.py:248); an int pins it (and lets the coarse graph be cached)
        self.__dict__["coarse_start"] = None
        self.device = device
        self.node_features_normalizer = Normalizer(1, node_in_dim, "node_features_normalizer", device)
        self.edge_features_normalizer = Normalizer(1, edge_in_dim, "edge_features_normalizer", device)
        self.output_normalizer = Normalizer(time_dim, node_out_dim, "output_normalizer", device)
        self.__dict__["_topo"] = TopologyCache()
        self.__dict__["_ws"] = None
        self.__dict__["halo"] = None          # set by gnn_amd.partition for mesh-partitioned runs

    # ----------------------------------------------------------------- plumbing
    def _node_feat_dim(self, graph) -> int:
        return 3 + graph["swelling_phi"].shape[1] + 5 + (2 if self.with_mat_params else 0)

    def _workspace(self, device, n, e, node_f) -> _Workspace:
        ws = self.__dict__.get("_ws")
        if ws is None or ws.key != (str(device), n, e, node_f):
            ws = _Workspace(device, n, e, node_f)
            self.__dict__["_ws"] = ws
        return ws

    def topology(self, graph) -> Topology:
        if "_topo" not in self.__dict__:
            self.__dict__["_topo"] = TopologyCache()
        return self.__dict__["_topo"].get(graph["edge_index"], graph["mesh_pos"].shape[0], graph["mesh_pos"].device)

    @staticmethod
    def _f32(t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(torch.float32).contiguous()

    def _packed(self, slot, blocks):
        pk = self.__dict__.get(slot)
        if pk is None:
            pk = self.__dict__[slot] = ops.PackedWeights()
        frozen = self.__dict__.get("_weights_frozen")
        if frozen is not None:                  # inside constant_weights(): convert once, on first use
            if slot not in frozen:
                pk.refresh(blocks)
                frozen.add(slot)
            return pk
        return pk.refresh(blocks)

    def constant_weights(self):
        """Context manager for a sequence of ``predict`` / ``forward`` calls during which the caller does not change
        the parameters (a rollout): the prepared weight sets are converted on the first call only instead of on
        every call."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            outer = self.__dict__.get("_weights_frozen")
            if outer is None:
                self.__dict__["_weights_frozen"] = set()
            try:
                yield self
            finally:
                if outer is None:
                    self.__dict__["_weights_frozen"] = None
        return scope()

    def _coarse_graph(self, graph):
        """Coarse graph of the frame's mesh (of every block of a batched frame, ``graph["batch_sizes"]``)."""
        from .coarse import CoarseGraph
        mesh_pos = graph["mesh_pos"]
        sizes = tuple(int(b) for b in graph["batch_sizes"]) if "batch_sizes" in graph else None
        start = self.__dict__.get("coarse_start")
        n = mesh_pos.shape[0]
        if start is None:                         # the reference's draw (core/meshgraphnet.py:248), same call on the same device
            starts = [int(torch.randint(0, b, (1,), device=mesh_pos.device)) for b in (sizes or (n,))]
            return CoarseGraph(mesh_pos, self.sample_ratio, starts, mesh_pos.device, sizes)
        key = (mesh_pos.data_ptr(), mesh_pos._version, n, str(mesh_pos.device), int(start), float(self.sample_ratio), sizes)
        hit = self.__dict__.get("_coarse")
        if hit is None or hit[0] != key:
            hit = (key, CoarseGraph(mesh_pos, self.sample_ratio, int(start), mesh_pos.device, sizes), mesh_pos)
            self.__dict__["_coarse"] = hit
        return hit[1]

    def _coarse_partition(self, halo, mesh_local):
        """Coarse graph of a partitioned mesh: the same on every rank, cut along the ranks' node ranges
        (``partition.CoarsePartition``).  The first sample of an unpinned model is rank 0's draw."""
        from .partition import CoarsePartition
        start = self.__dict__.get("coarse_start")
        if start is None:
            draw = torch.zeros(1, dtype=torch.float64, device=mesh_local.device)
            if halo.rank == 0:
                draw += float(torch.randint(0, halo.global_nodes, (1,)))
            return CoarsePartition(halo, mesh_local, self.sample_ratio, int(halo.all_reduce_(draw)))
        key = (id(halo), mesh_local.data_ptr(), mesh_local._version, int(start), float(self.sample_ratio))
        hit = self.__dict__.get("_coarse")
        if hit is None or hit[0] != key:
            hit = (key, CoarsePartition(halo, mesh_local, self.sample_ratio, int(start)), mesh_local, halo)
            self.__dict__["_coarse"] = hit
        return hit[1]

    def _coarse_workspace(self, device, cg, n_own):
        nc, ec = cg.num_nodes, cg.topology.num_edges
        cw = self.__dict__.get("_cws")
        if cw is None or cw.key != (str(device), n_own, nc, ec):
            f32 = dict(dtype=torch.float32, device=device)
            cw = type("CoarseWorkspace", (), {})()
            cw.key = (str(device), n_own, nc, ec)
            cw.x, cw.ps, cw.pr, cw.aggr = (torch.empty(nc, H, **f32) for _ in range(4))
            cw.e, cw.edge_feat = torch.empty(ec, H, **f32), torch.empty(ec, 7, **f32)
            cw.map_back, cw.hidden = torch.empty(n_own, H, **f32), torch.empty(n_own, H, **f32)
            self.__dict__["_cws"] = cw
        return cw

    def _decode(self, ws, n, **kw):
        """Decoder (+ integrator): over the fine latents, or over cat[fine, coarse mapped back] in the multiscale model."""
        if self.sample_ratio > 0:
            cw, d = self.__dict__["_cws"], self.node_decoder
            ops.decode_cat(ws.x[:n], cw.map_back, d[0].weight.data, d[0].bias.data, d[2].weight.data, d[2].bias.data,
                           self.time_dim, self.node_out_dim, cw.hidden, **kw)
        else:
            ops.decode(ws.x[:n], self.node_decoder.weights(), self.time_dim, self.node_out_dim, **kw)

    def _encode_process(self, graph):
        """Features -> normalise (accumulating) -> encode -> process.  Returns the workspace
        holding the final node latents (core/meshgraphnet.py:175-185)."""
        dev = graph["mesh_pos"].device
        if dev.type != "cuda":
            raise RuntimeError("EncodeProcessDecodeMultiScale runs on the GPU only: move the graph with graph.to('cuda')")
        if next(self.parameters()).device != dev:
            raise RuntimeError(f"model parameters are on {next(self.parameters()).device}, graph on {dev}")
        halo = self.__dict__.get("halo")
        topo = halo.topology if halo is not None else self.topology(graph)
        n, e = topo.num_nodes, topo.num_edges
        node_f = self._node_feat_dim(graph)
        if node_f != self.node_encoder[0].weight.shape[1]:
            raise RuntimeError(f"node features have width {node_f} but node_in_dim is {self.node_encoder[0].weight.shape[1]}")
        ws = self._workspace(dev, n, e, node_f)
        world, mesh, phi = self._f32(graph["world_pos"]), self._f32(graph["mesh_pos"]), self._f32(graph["phi"])
        n_own = halo.num_owned if halo is not None else n
        if halo is not None:
            if mesh.shape[0] != n:
                raise RuntimeError(f"partitioned run: mesh_pos must hold the {n} local rows [owned | halo]")
            if world.shape[0] < n_own or phi.shape[0] < n_own:
                raise RuntimeError(f"partitioned run: world_pos / phi must hold at least the {n_own} owned rows")
            world, phi = halo.exchange_state(world, phi)      # [owned | halo] buffers of the plan, halo rows refreshed
        mat = self._f32(graph["mat_param"]) if self.with_mat_params else None
        nn_, en_ = self.node_features_normalizer, self.edge_features_normalizer
        # on one GPU a live normaliser's column sums are formed by the feature kernel itself and folded by one more launch
        pn = nn_.partials_for(n_own) if halo is None and nn_.accumulating() and n_own > 0 else None
        pe = en_.partials_for(e) if halo is None and en_.accumulating() and e > 0 else None
        ops.node_features(world[:n_own], mesh[:n_own], phi[:n_own], self._f32(graph["swelling_phi"])[:n_own],
                          self._f32(graph["node_type"])[:n_own], mat, ws.node_feat[:n_own], partials=pn)
        ops.edge_features(world, mesh, phi, topo.sender, topo.recv, ws.edge_feat, partials=pe)
        if halo is None:
            ms_n = nn_.fold_partials(n_own) if pn is not None else nn_.stats_for(ws.node_feat, True)
            ms_e = en_.fold_partials(e) if pe is not None else en_.stats_for(ws.edge_feat, True)
        else:                                     # partial sums of both normalisers in one all-reduce
            ms_n, ms_e = halo.feature_stats(self.node_features_normalizer, ws.node_feat[:n_own],
                                            self.edge_features_normalizer, ws.edge_feat)
        ops.encode(ws.node_feat[:n_own], ms_n, self.node_encoder.weights(), ws.x[:n_own])
        ops.encode(ws.edge_feat, ms_e, self.edge_encoder.weights(), ws.e)
        cg = cw = chalo = None
        if self.sample_ratio > 0:                 # core/meshgraphnet.py:179-182
            if halo is not None:                  # the coarse graph has its own plan over the ranks' node ranges
                cg = self._coarse_partition(halo, graph["mesh_pos"])
                chalo = cg.plan
                (cworld, cphi), cmesh = cg.state(world, phi), cg.mesh
                c_snd, c_rcv, c_first, nc_own = cg.topology.sender, cg.topology.recv, cg.first_local, cg.num_owned
            else:                                 # end points addressed by their mesh node ids
                cg = self._coarse_graph(graph)
                cworld, cphi, cmesh = world, phi, mesh
                c_snd, c_rcv, c_first, nc_own = cg.sender_orig, cg.recv_orig, cg.first, cg.num_nodes
            cw = self._coarse_workspace(dev, cg, n_own)
            ops.edge_features(cworld, cmesh, cphi, c_snd, c_rcv, cw.edge_feat)
            ms_c = self.coarse_edge_features_normalizer.stats_for(cw.edge_feat, True,
                                                                  chalo.sum_hook("edge") if chalo is not None else None)
            ops.encode(cw.edge_feat, ms_c, self.coarse_edge_encoder.weights(), cw.e)
            ops.gather_rows(ws.x, c_first, cw.x[:nc_own])  # encoder output of the sampled nodes (:182)
        blocks = list(self.processors)
        # prepared weight sets (one conversion launch per call: a snapshot of the parameters, so never stale)
        use_pk = ops.matmul_mode() == "f16x2" and os.environ.get("MGN_PACKED_WEIGHTS", "1") != "0"
        pk = self._packed("_pk", blocks) if use_pk and blocks else None
        for i, proc in enumerate(blocks):
            last = i + 1 == len(blocks)           # nothing reads the edge latents after the last block
            proc.step_csr(ws.x, ws.e, topo, ws.ps, ws.pr, ws.aggr, halo, projected=i > 0,
                          next_block=None if last else blocks[i + 1], store_edges=not last,
                          packed=(pk.edge[i], pk.node[i]) if pk is not None else None)
        if cg is not None:                        # :187-194
            cblocks = list(self.coarse_processors)
            cpk = self._packed("_cpk", cblocks) if use_pk and cblocks else None
            for i, proc in enumerate(cblocks):
                last = i + 1 == len(cblocks)
                proc.step_csr(cw.x, cw.e, cg.topology, cw.ps, cw.pr, cw.aggr, chalo, projected=i > 0,
                              next_block=None if last else cblocks[i + 1], store_edges=not last,
                              packed=(cpk.edge[i], cpk.node[i]) if cpk is not None else None)
            cw.map_back.zero_()
            ops.scatter_rows(cw.x[:nc_own], c_first, cw.map_back)
        return ws, topo

    # ----------------------------------------------------------------- reference API
    @_on_graph_device
    def forward(self, batch, check_range: bool = True):
        """Normalised per-step deltas [time_dim, N, node_out_dim] (core/meshgraphnet.py:168-211).
        ``check_range`` as in :meth:`predict` (never checked while the stream is being captured into a HIP graph)."""
        ws, topo = self._encode_process(batch)
        halo = self.__dict__.get("halo")
        n = halo.num_owned if halo is not None else topo.num_nodes
        delta = torch.empty(self.time_dim, n, self.node_out_dim, dtype=torch.float32, device=ws.x.device)
        self._decode(ws, n, delta_norm=delta)
        if check_range and not torch.cuda.is_current_stream_capturing():
            ops.raise_if_out_of_range("forward")
        return delta

    @_on_graph_device
    def predict(self, graph, check_range: bool = True, feedback: bool = False):
        """Next state(s) [time_dim, N, 3]: decode, de-normalise, zero the deltas of
        Dirichlet nodes, add to the current state (core/meshgraphnet.py:305-317).

        ``feedback``: the decoder launch also writes the last predicted step into ``graph["world_pos"]`` /
        ``graph["phi"]`` themselves (float32, contiguous device tensors) -- the state feedback of the rollout loop
        (core/rollout.py:29-32) without separate copies.  Plain model on one GPU only.

        ``check_range``: read the range flag of the default matrix path after the launches (one small
        device-to-host copy, which waits for the step) and raise if an operand left the fp16 range; loops
        that check once at their end (``rollout``) pass False.  The flag is sticky: a check that comes later
        loses nothing.  Never checked while the stream is being captured into a HIP graph."""
        if self.node_out_dim != 3:
            raise RuntimeError("predict: node_out_dim must be 3 (world_pos x, y and phi)")
        with torch.no_grad():
            ws, topo = self._encode_process(graph)
            halo = self.__dict__.get("halo")
            n = halo.num_owned if halo is not None else topo.num_nodes
            ms_o = self.output_normalizer.stats_for(None, accumulate=False)
            pred = torch.empty(self.time_dim, n, 3, dtype=torch.float32, device=ws.x.device)
            world, phi = self._f32(graph["world_pos"])[:n], self._f32(graph["phi"])[:n]
            kw = {}
            if feedback:
                if halo is not None or self.sample_ratio > 0 or world.data_ptr() != graph["world_pos"].data_ptr() \
                        or phi.data_ptr() != graph["phi"].data_ptr():
                    raise RuntimeError("predict(feedback=True): plain model on one GPU, state tensors float32 and contiguous")
                kw["next_state"] = (world, phi)
            self._decode(ws, n, out_mean_std=ms_o, world_pos=world, phi=phi, node_type=self._f32(graph["node_type"])[:n],
                         pred=pred, **kw)
        if check_range and not torch.cuda.is_current_stream_capturing():
            ops.raise_if_out_of_range("predict")
        return pred

    def _normalized_target(self, graph) -> torch.Tensor:
        """``output_normalizer(target - current state)`` [time_dim, N, 3], accumulating
        (core/meshgraphnet.py:286-290)."""
        target = self._f32(graph["target"])
        cur = torch.cat([self._f32(graph["world_pos"]), self._f32(graph["phi"])], dim=-1)
        halo = self.__dict__.get("halo")
        if halo is None:
            return self.output_normalizer(target - cur).contiguous()
        n = halo.num_owned             # owned rows only; the statistics see the sums over all ranks
        return self.output_normalizer((target[:, :n] - cur[:n]).contiguous(),
                                      partial_sums_hook=halo.sum_hook("node")).contiguous()

    def _param_list(self):
        """(names, parameters) in ``named_parameters`` order, listed once: walking the module tree
        costs more host time than a small mesh's whole forward pass."""
        cached = self.__dict__.get("_mgn_params")
        if cached is None:
            named = list(self.named_parameters())
            cached = ([k for k, _ in named], [p for _, p in named])
            self.__dict__["_mgn_params"] = cached
        return cached

    @_on_graph_device
    def loss(self, graph, check_range: bool = True):
        """Masked squared error between the normalised target delta and the prediction
        (core/meshgraphnet.py:285-304): returns (total, ux, uy, phi).  The output normaliser
        accumulates first (:290).  With gradients enabled the result carries the backward pass
        of the HIP training kernels, so ``loss.backward()`` fills ``.grad`` of every parameter
        as run_train_multiscale.py:139-141 expects.  ``check_range``: as in :meth:`predict` for a call without
        gradients; a training call does not wait for the GPU -- it polls the (sticky) flag without blocking, so an
        operand beyond the range, in this forward pass or the backward pass that follows, raises at the next call
        (``ops.raise_if_out_of_range`` after the loop catches the last step)."""
        from . import training
        tgt = self._normalized_target(graph)
        training_call = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list()[1])
        if training_call:
            comps = training.loss_components(self, graph, tgt)
        else:
            comps, _ = training.forward_saving(self, graph, tgt)
        if check_range and training_call:
            ops.poll_out_of_range("loss")
        elif check_range and not torch.cuda.is_current_stream_capturing():
            ops.raise_if_out_of_range("loss")
        return comps[0] + comps[1] + comps[2], comps[0], comps[1], comps[2]

    def save_model(self, path):
        """Checkpoint layout of the reference (core/meshgraphnet.py:318-324)."""
        torch.save(self.state_dict(), os.path.join(path, "model_weights.pth"))
        torch.save(self.output_normalizer, os.path.join(path, "output_normalizer.pth"))
        torch.save(self.node_features_normalizer, os.path.join(path, "node_features_normalizer.pth"))
        torch.save(self.edge_features_normalizer, os.path.join(path, "edge_features_normalizer.pth"))
        if self.sample_ratio > 0:
            torch.save(self.coarse_edge_features_normalizer, os.path.join(path, "coarse_edge_features_normalizer.pth"))

    def load_model(self, path):
        """Loads the reference's checkpoint directory (core/meshgraphnet.py:326-332).  The
        weights and the three pickled normalisers are read with the tensor-only unpickler
        and mapped onto this model's device."""
        dev = next(self.parameters()).device
        self.load_state_dict(torch.load(os.path.join(path, "model_weights.pth"), weights_only=True, map_location=dev))
        self.output_normalizer = load_normalizer(os.path.join(path, "output_normalizer.pth"), dev)
        self.node_features_normalizer = load_normalizer(os.path.join(path, "node_features_normalizer.pth"), dev)
        self.edge_features_normalizer = load_normalizer(os.path.join(path, "edge_features_normalizer.pth"), dev)
        if self.sample_ratio > 0:
            self.coarse_edge_features_normalizer = load_normalizer(os.path.join(path, "coarse_edge_features_normalizer.pth"), dev)

    def _apply(self, fn, *args, **kwargs):
        """``.to(device)`` also moves the normaliser statistics, which the reference keeps
        as plain attributes created on the constructor's ``device``."""
        super()._apply(fn, *args, **kwargs)
        try:
            dev = next(self.parameters()).device
        except StopIteration:
            return self
        norms = [self.node_features_normalizer, self.edge_features_normalizer, self.output_normalizer]
        if self.sample_ratio > 0:
            norms.append(self.coarse_edge_features_normalizer)
        for nrm in norms:
            if nrm._acc_sum.device != dev:
                nrm.to_device(dev)
        self.__dict__["_ws"] = None
        self.__dict__["_cws"] = None
        self.__dict__["_coarse"] = None
        self.__dict__["_mgn_params"] = None
        return self
