"""Training step on the HIP kernels: ``loss(graph)`` with gradients for every parameter.

Covers what run_train_multiscale.py:139-141 runs per batch -- ``model.loss(batch)`` then
``loss.backward()`` -- for the reference's forward (core/meshgraphnet.py:168-211) and loss
(:285-304).  The forward keeps each layer's inputs and the two hidden activations of every
MLP; the backward walks the layers in reverse with the blocks of ``mgn_train.hip``.  Weight
gradients are deterministic (no atomics).  ``torch.autograd`` only carries the three loss
components out and the parameter gradients back; no arithmetic of the model runs in torch.
"""
from __future__ import annotations

import os
import threading
from typing import Dict, List

import torch

from . import ops
from ._lib import H


def _empty(rows, dev):
    return torch.empty(rows, H, dtype=torch.float32, device=dev)


class _Scratch:
    """Partial-sum buffer of the weight-gradient kernels: one per device and host thread (two
    threads driving the same GPU must not share it)."""
    _by_dev: Dict[tuple, torch.Tensor] = {}

    @classmethod
    def get(cls, dev) -> torch.Tensor:
        key = (str(dev), threading.get_ident())
        buf = cls._by_dev.get(key)
        if buf is None:
            buf = torch.empty(ops.train_scratch_bytes(), dtype=torch.uint8, device=dev)
            cls._by_dev[key] = buf
        return buf


class _Deferred:
    """One :class:`ops.DeferredReductions` per device and host thread, kept from step to step."""
    _by_dev: Dict[tuple, "ops.DeferredReductions"] = {}

    @classmethod
    def get(cls, dev):
        key = (str(dev), threading.get_ident())
        d = cls._by_dev.get(key)
        if d is None:
            d = cls._by_dev[key] = ops.DeferredReductions(dev)
        return d


def _pad_cols(w: torch.Tensor) -> torch.Tensor:
    """[128, F] -> [128, 128] with zero columns (first Linear of an encoder)."""
    out = torch.zeros(H, H, dtype=torch.float32, device=w.device)
    out[:, : w.shape[1]].copy_(w.detach())
    return out


def _pad_rows(w: torch.Tensor, b: torch.Tensor):
    """[O, 128], [O] -> [128, 128], [128] with zero rows (last Linear of the decoder)."""
    wp = torch.zeros(H, H, dtype=torch.float32, device=w.device)
    bp = torch.zeros(H, dtype=torch.float32, device=w.device)
    wp[: w.shape[0]].copy_(w.detach())
    bp[: b.shape[0]].copy_(b.detach())
    return wp, bp


def forward_saving(model, graph, target_norm: torch.Tensor, fused=None, keep_hidden: bool = False):
    """Runs the forward pass with the training kernels.  Returns (loss3 [3], saved) where
    ``saved`` holds what ``backward`` needs.

    ``fused`` (default: on for the default matrix path): the edge update of every block runs in the
    inference kernel and keeps only its input; the backward recomputes the two hidden activations
    tile by tile (``mgn_train_edge_bwd``).  Otherwise every Linear / LayerNorm is its own launch
    and both hidden activations of every edge MLP are kept.  ``keep_hidden`` (diagnostics): the fused
    path also evaluates and keeps the two hidden activations, from the same inputs and with the same
    arithmetic as the fused kernels, so that a caller can read the ReLU pattern of the step."""
    if fused is None:
        fused = ops.matmul_mode() == "f16x2" and os.environ.get("MGN_TRAIN_FUSED", "1") != "0"
    halo = model.__dict__.get("halo")
    dev = graph["mesh_pos"].device
    if dev.type != "cuda":
        raise RuntimeError("the training kernels run on the GPU only")
    if model.node_out_dim != 3:
        raise RuntimeError("loss: node_out_dim must be 3")
    f32 = model._f32
    topo = halo.topology if halo is not None else model.topology(graph)
    n_loc, e, T = topo.num_nodes, topo.num_edges, model.time_dim
    n = halo.num_owned if halo is not None else n_loc          # rows this rank computes; [n, n_loc) is the halo
    scratch = _Scratch.get(dev)
    node_f = model._node_feat_dim(graph)
    if node_f != model.node_encoder[0].weight.shape[1]:
        raise RuntimeError(f"node features have width {node_f} but node_in_dim is {model.node_encoder[0].weight.shape[1]}")

    # features and normalisers (accumulating), as in the inference path
    world, mesh, phi = f32(graph["world_pos"]), f32(graph["mesh_pos"]), f32(graph["phi"])
    node_type = f32(graph["node_type"])[:n].contiguous()
    if halo is not None:
        if mesh.shape[0] != n_loc:
            raise RuntimeError(f"partitioned run: mesh_pos must hold the {n_loc} local rows [owned | halo]")
        world, phi = halo.exchange_state(world, phi)      # [owned | halo] buffers of the plan, halo rows refreshed
    mat = f32(graph["mat_param"]) if model.with_mat_params else None
    nfeat = torch.empty(n, node_f, dtype=torch.float32, device=dev)
    efeat = torch.empty(e, 7, dtype=torch.float32, device=dev)
    ops.node_features(world[:n], mesh[:n], phi[:n], f32(graph["swelling_phi"])[:n], node_type, mat, nfeat)
    ops.edge_features(world, mesh, phi, topo.sender, topo.recv, efeat)
    if halo is None:
        ms_n = model.node_features_normalizer.stats_for(nfeat, True)
        ms_e = model.edge_features_normalizer.stats_for(efeat, True)
    else:                                          # partial sums of both normalisers in one all-reduce
        ms_n, ms_e = halo.feature_stats(model.node_features_normalizer, nfeat, model.edge_features_normalizer, efeat)
    nf_pad, ef_pad = _empty(n, dev), _empty(e, dev)
    ops.train_pad_features(nfeat, ms_n, nf_pad)
    ops.train_pad_features(efeat, ms_e, ef_pad)
    del nfeat, efeat

    def encoder(seq, inp):
        w0p = _pad_cols(seq[0].weight)
        h1, v, y = _empty(inp.shape[0], dev), _empty(inp.shape[0], dev), _empty(inp.shape[0], dev)
        ops.train_linear_fwd(inp, w0p, 0, h1, bias=seq[0].bias.detach(), relu=True)
        ops.train_linear_fwd(h1, seq[2].weight.detach(), 0, v, bias=seq[2].bias.detach())
        ops.train_ln_fwd(v, seq[3].weight.detach(), seq[3].bias.detach(), y=y)
        return y, (inp, h1, v)

    x, enc_n = encoder(model.node_encoder, nf_pad)
    el, enc_e = encoder(model.edge_encoder, ef_pad)

    def run_blocks(blocks, x, el, topo, n, n_loc, e, halo):
        """A stack of processor blocks on one topology; returns the final node latents and what the backward needs."""
        layers: List[tuple] = []
        ps, pr, eprime = _empty(n_loc, dev), _empty(n_loc if fused else n, dev), (None if fused else _empty(e, dev))
        rowptr = topo.rowptr[: n + 1]               # halo nodes receive nothing here: their rows stay with the owners
        for i, blk in enumerate(blocks):
            em, nm = blk.edge_mlp, blk.node_mlp
            w0 = em[0].weight.detach()
            if fused and i > 0:
                ps, pr = _empty(n_loc, dev), _empty(n_loc, dev)      # every block's projections are kept for the backward
            ops.train_linear_fwd(x, w0, 0, ps[:n])
            if halo is not None:
                pending = halo.exchange_rows(ps, async_op=True)      # sender projections of the halo nodes
            ops.train_linear_fwd(x, w0, H, pr[:n], bias=em[0].bias.detach())
            if halo is not None and pending is not None:
                pending.wait()
            last = i + 1 == len(blocks)
            att = None
            if blk.attention:
                # message weights from the block's input node latents (core/meshgraphnet.py:100-107)
                a_proj = _empty(n_loc, dev)
                alpha, wts = torch.empty(e, dtype=torch.float32, device=dev), torch.empty(e, dtype=torch.float32, device=dev)
                ops.train_linear_fwd(x, blk.attn_lin.weight.detach(), 0, a_proj[:n], bias=blk.attn_lin.bias.detach())
                if halo is not None:
                    halo.exchange_rows(a_proj)                       # projected rows of the halo senders
                ops.edge_attention(a_proj, topo.rowptr, topo.sender, topo.recv, alpha, wts)
                att = (a_proj, alpha, wts)
            if fused:
                # whole edge update in one kernel; hidden activations are recomputed in the backward
                # the edge kernel writes the sum of every receiver that has in-edges; the others must read as zero
                aggr = _empty(n, dev) if topo.min_in_degree(n) > 0 else torch.zeros(n, H, dtype=torch.float32, device=dev)
                # the edge latents after the last block are unused, except by the backward of the message weights
                e_next = None if last and att is None else _empty(e, dev)
                with ops.timed("mp_edge"):
                    ops.mp_edge(el, ps, pr, topo.rowptr, topo.sender, topo.recv, em.weights(), e_next, aggr,
                                edge_weight=None if att is None else att[2])
                h1e = ve = None
                if keep_hidden:
                    h1e, ve = _empty(e, dev), _empty(e, dev)
                    ops.train_linear_fwd(el, w0, 2 * H, h1e, p=ps, idx_p=topo.sender, q=pr, idx_q=topo.recv, relu=True)
                    ops.train_linear_fwd(h1e, em[2].weight.detach(), 0, ve, bias=em[2].bias.detach(), relu=True)
            else:
                h1e, ve = _empty(e, dev), _empty(e, dev)
                ops.train_linear_fwd(el, w0, 2 * H, h1e, p=ps, idx_p=topo.sender, q=pr, idx_q=topo.recv, relu=True)
                ops.train_linear_fwd(h1e, em[2].weight.detach(), 0, ve, bias=em[2].bias.detach(), relu=True)
                e_next = None if last else _empty(e, dev)      # the edge latents after the last block are unused
                if att is not None:
                    eprime = _empty(e, dev)                    # the messages themselves are kept for the weights' backward
                ops.train_ln_fwd(ve, em[4].weight.detach(), em[4].bias.detach(), y=eprime,
                                 res=None if last else el, yres=e_next)
                aggr = _empty(n, dev)
                if att is None:
                    ops.segment_sum(eprime, rowptr, aggr)
                else:
                    ops.segment_sum_weighted(eprime, rowptr, att[2], aggr)
            h1n, vn, x_next = _empty(n, dev), _empty(n, dev), _empty(n, dev)
            ops.train_linear_fwd(x, nm[0].weight.detach(), 0, h1n, b=aggr, koff_b=H, bias=nm[0].bias.detach(), relu=True)
            ops.train_linear_fwd(h1n, nm[2].weight.detach(), 0, vn, bias=nm[2].bias.detach(), relu=True)
            ops.train_ln_fwd(vn, nm[4].weight.detach(), nm[4].bias.detach(), res=x, yres=x_next)
            # with attention: the projected rows, alpha, the weights and the messages (fused path: as e_next - el)
            layers.append((x, el, h1e, ve, aggr, h1n, vn, ps if fused else None, pr if fused else None,
                           None if att is None else att + ((e_next, el) if fused else (eprime, None))))
            x, el = x_next, e_next

        return x, layers

    coarse = None
    if model.sample_ratio > 0:
        # coarse branch (core/meshgraphnet.py:179-194): its edge latents from the fine fields of the coarse end points,
        # its node latents = the encoder output of the sampled nodes
        if halo is not None:                       # partitioned: the coarse graph runs on its own plan
            cg = model._coarse_partition(halo, graph["mesh_pos"])
            chalo = cg.plan
            (cworld, cphi), cmesh = cg.state(world, phi), cg.mesh
            c_snd, c_rcv, c_first, nc, nc_loc = cg.topology.sender, cg.topology.recv, cg.first_local, cg.num_owned, cg.num_nodes
        else:
            cg, chalo = model._coarse_graph(graph), None
            cworld, cphi, cmesh = world, phi, mesh
            c_snd, c_rcv, c_first, nc, nc_loc = cg.sender_orig, cg.recv_orig, cg.first, cg.num_nodes, cg.num_nodes
        ec = cg.topology.num_edges
        efeat_c = torch.empty(ec, 7, dtype=torch.float32, device=dev)
        ops.edge_features(cworld, cmesh, cphi, c_snd, c_rcv, efeat_c)
        ms_c = model.coarse_edge_features_normalizer.stats_for(efeat_c, True, chalo.sum_hook("edge") if chalo is not None else None)
        ecf_pad = _empty(ec, dev)
        ops.train_pad_features(efeat_c, ms_c, ecf_pad)
        elc, enc_c = encoder(model.coarse_edge_encoder, ecf_pad)
        xc = _empty(nc, dev)
        ops.gather_rows(x, c_first, xc)
    x, layers = run_blocks(list(model.processors), x, el, topo, n, n_loc, e, halo)
    map_back = None
    if model.sample_ratio > 0:
        xc, layers_c = run_blocks(list(model.coarse_processors), xc, elc, cg.topology, nc, nc_loc, ec, chalo)
        map_back = torch.zeros(n, H, dtype=torch.float32, device=dev)
        ops.scatter_rows(xc, c_first, map_back)                      # :192-193
        coarse = dict(cg=cg, enc_c=enc_c, layers=layers_c, halo=chalo, first=c_first, nc=nc, nc_loc=nc_loc, ec=ec)

    dec = model.node_decoder
    hd, decoded = _empty(n, dev), _empty(n, dev)
    if map_back is None:
        ops.train_linear_fwd(x, dec[0].weight.detach(), 0, hd, bias=dec[0].bias.detach(), relu=True)
    else:                                           # first Linear over cat[fine, coarse mapped back] (:194, :204)
        ops.train_linear_fwd(x, dec[0].weight.detach(), 0, hd, b=map_back, koff_b=H, bias=dec[0].bias.detach(), relu=True)
    w2p, b2p = _pad_rows(dec[2].weight, dec[2].bias)
    ops.train_linear_fwd(hd, w2p, 0, decoded, bias=b2p)
    loss3 = torch.empty(3, dtype=torch.float32, device=dev)
    ops.train_loss_fwd(decoded, target_norm, node_type, T, loss3, scratch)
    if halo is not None:
        halo.all_reduce_(loss3)                     # every term of the loss is a sum over nodes
    saved = dict(topo=topo, halo=halo, fused=fused, enc_n=enc_n, enc_e=enc_e, layers=layers, x_last=x, hd=hd, decoded=decoded, w2p=w2p,
                 target_norm=target_norm, node_type=node_type, coarse=coarse, map_back=map_back)
    return loss3, saved


def backward(model, saved, grad3: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Gradients of ``sum_c grad3[c] * loss3[c]`` for every parameter, keyed like ``named_parameters``.
    The per-workgroup partial sums of all weight gradients are reduced by one launch at the end of the pass
    (``MGN_TRAIN_DEFER=0``: one small launch per call, the same sums bit for bit)."""
    dev = grad3.device
    if os.environ.get("MGN_TRAIN_DEFER", "1") == "0":
        return _backward(model, saved, grad3, _Scratch.get(dev), None)
    deferred = _Deferred.get(dev)
    with deferred:
        return _backward(model, saved, grad3, deferred, deferred.flush)


def _backward(model, saved, grad3, scratch, flush) -> Dict[str, torch.Tensor]:
    topo, halo = saved["topo"], saved["halo"]
    dev = grad3.device
    n_loc, e, T = topo.num_nodes, topo.num_edges, model.time_dim
    n = halo.num_owned if halo is not None else n_loc
    grads: Dict[str, torch.Tensor] = {}
    narrowed = []                                 # (name, padded gradient, slicer): cut to the parameter's shape once it is reduced

    def g_like(p):
        return torch.empty_like(p, dtype=torch.float32)

    def vec():
        return torch.empty(H, dtype=torch.float32, device=dev)

    # ---- decoder
    dec = model.node_decoder
    n_out = dec[2].weight.shape[0]
    gdec = _empty(n, dev)
    ops.train_loss_bwd(saved["decoded"], saved["target_norm"], saved["node_type"], grad3.contiguous(), T, gdec)
    dw2p, db2p = torch.empty(H, H, dtype=torch.float32, device=dev), vec()
    gh = _empty(n, dev)
    ops.train_linear_bwd(gdec, scratch, a=saved["hd"], w=saved["w2p"], mask=saved["hd"], ga=gh, dw=dw2p, db=db2p)
    narrowed.append(("node_decoder.2.weight", dw2p, lambda t: t[:n_out].contiguous()))
    narrowed.append(("node_decoder.2.bias", db2p, lambda t: t[:n_out].contiguous()))
    grads["node_decoder.2.weight"] = grads["node_decoder.2.bias"] = None          # keeps the parameter order of the dict
    gx = gdec                                     # reuse: gradient wrt the node latents entering the decoder
    dw, db = g_like(dec[0].weight), vec()
    coarse, gmap = saved.get("coarse"), None
    if coarse is None:
        ops.train_linear_bwd(gh, scratch, a=saved["x_last"], w=dec[0].weight.detach(), ga=gx, dw=dw, db=db)
    else:                                         # first Linear over cat[fine, coarse mapped back]: two column blocks
        gmap = _empty(n, dev)
        wd0 = dec[0].weight.detach()
        ops.train_linear_bwd(gh, scratch, a=saved["x_last"], w=wd0, koff=0, ga=gx, dw=dw, dw_koff=0, db=db)
        ops.train_linear_bwd(gh, scratch, a=saved["map_back"], w=wd0, koff=H, ga=gmap, dw=dw, dw_koff=H)
    grads["node_decoder.0.weight"], grads["node_decoder.0.bias"] = dw, db

    def blocks_backward(blocks, prefix, layers, topo, halo, gx, n, n_loc, e):
        """Processor blocks in reverse: ``gx`` (gradient wrt the node latents leaving the stack) becomes the gradient wrt
        those entering it, in place; returns the gradient wrt the edge latents entering it (None without blocks)."""
        ge = None                                     # gradient wrt the edge latents leaving the block
        g2n, g1n, gaggr, dpr = (_empty(n, dev) for _ in range(4))
        dps_loc = _empty(n_loc, dev)
        dps = dps_loc[:n]
        g2e, g1e = (None if saved["fused"] else _empty(e, dev)), _empty(e, dev)
        s_rowptr, s_order = topo.sender_view()
        rowptr = topo.rowptr[: n + 1]
        for i in range(len(blocks) - 1, -1, -1):
            blk, pre = blocks[i], f"{prefix}.{i}."
            em, nm = blk.edge_mlp, blk.node_mlp
            x, el, h1e, ve, aggr, h1n, vn, ps_b, pr_b, att = layers[i]
            # node MLP: x_next = x + LN(ReLU(V2 ReLU(V1 [x, aggr] + c1) + c2))
            dg, dbt = vec(), vec()
            ops.train_ln_bwd(vn, nm[4].weight.detach(), g2n, dg, dbt, scratch, g=gx, relu_mask=True)
            grads[pre + "node_mlp.4.weight"], grads[pre + "node_mlp.4.bias"] = dg, dbt
            dw, db = g_like(nm[2].weight), vec()
            ops.train_linear_bwd(g2n, scratch, a=h1n, w=nm[2].weight.detach(), mask=h1n, ga=g1n, dw=dw, db=db)
            grads[pre + "node_mlp.2.weight"], grads[pre + "node_mlp.2.bias"] = dw, db
            dw, db = g_like(nm[0].weight), vec()
            w_n0 = nm[0].weight.detach()
            ops.train_linear_bwd(g1n, scratch, a=x, w=w_n0, koff=0, res=gx, ga=gx, dw=dw, dw_koff=0, db=db)
            ops.train_linear_bwd(g1n, scratch, a=aggr, w=w_n0, koff=H, ga=gaggr, dw=dw, dw_koff=H)
            grads[pre + "node_mlp.0.weight"], grads[pre + "node_mlp.0.bias"] = dw, db
            # edge MLP: e' = LN(ReLU(W2 ReLU(ps[s] + pr[r] + W0e e) + b2)); upstream = ge + gaggr[receiver]
            dg, dbt = vec(), vec()
            dw, db = g_like(em[2].weight), vec()
            if att is not None:
                # aggr = sum of w_e * m_e with m_e = e_next - el: the weights' gradient goes back to attn_lin and the
                # node latents; m_e receives w_e * gaggr[r_e] on top of what reaches e_next
                a_proj, alpha, wts, m_out, m_in = att
                gw, gdot = torch.empty(e, dtype=torch.float32, device=dev), torch.empty(e, dtype=torch.float32, device=dev)
                ge_m, ga_loc = _empty(e, dev), _empty(n_loc, dev)
                ops.train_attn_edge_bwd(gaggr, topo.recv, m_out, m_in, wts, ge, gw, ge_m)
                ops.train_attn_alpha_bwd(rowptr, alpha, gw, gdot)
                ops.train_attn_node_grad(a_proj, rowptr, None, topo.sender, gdot, ga_loc[:n], False)     # as receivers (owned)
                if n_loc > n:
                    ga_loc[n:].zero_()
                ops.train_attn_node_grad(a_proj, s_rowptr, s_order, topo.recv, gdot, ga_loc, True)      # as senders (owned + halo)
                if halo is not None:
                    halo.return_rows(ga_loc)                         # the halo senders' sums go back to their owners
                dwa, dba = g_like(blk.attn_lin.weight), vec()
                ops.train_linear_bwd(ga_loc[:n], scratch, a=x, w=blk.attn_lin.weight.detach(), res=gx, ga=gx, dw=dwa, db=dba)
                grads[pre + "attn_lin.weight"], grads[pre + "attn_lin.bias"] = dwa, dba
                if saved["fused"]:
                    ops.train_edge_bwd(el, ge_m, torch.zeros_like(gaggr), ps_b, pr_b, topo.sender, topo.recv, rowptr,
                                       em[0].weight.detach(), em[2].weight.detach(), em[2].bias.detach(), em[4].weight.detach(),
                                       g1e, dpr, dw, db, dg, dbt, scratch)
                else:
                    ops.train_ln_bwd(ve, em[4].weight.detach(), g2e, dg, dbt, scratch, g=ge_m, relu_mask=True)
                    ops.train_linear_bwd(g2e, scratch, a=h1e, w=em[2].weight.detach(), mask=h1e, ga=g1e, dw=dw, db=db)
            elif saved["fused"]:
                ops.train_edge_bwd(el, ge, gaggr, ps_b, pr_b, topo.sender, topo.recv, rowptr, em[0].weight.detach(),
                                   em[2].weight.detach(), em[2].bias.detach(), em[4].weight.detach(), g1e, dpr, dw, db, dg, dbt,
                                   scratch)
            else:
                ops.train_ln_bwd(ve, em[4].weight.detach(), g2e, dg, dbt, scratch, g=ge, gg=gaggr, idx=topo.recv, relu_mask=True)
                ops.train_linear_bwd(g2e, scratch, a=h1e, w=em[2].weight.detach(), mask=h1e, ga=g1e, dw=dw, db=db)
            grads[pre + "edge_mlp.4.weight"], grads[pre + "edge_mlp.4.bias"] = dg, dbt
            grads[pre + "edge_mlp.2.weight"], grads[pre + "edge_mlp.2.bias"] = dw, db
            dw0, db0 = g_like(em[0].weight), vec()
            w_e0 = em[0].weight.detach()
            if ge is None:
                ge = _empty(e, dev)
                ops.train_linear_bwd(g1e, scratch, a=el, w=w_e0, koff=2 * H, ga=ge, dw=dw0, dw_koff=2 * H)
            else:
                ops.train_linear_bwd(g1e, scratch, a=el, w=w_e0, koff=2 * H, res=ge, ga=ge, dw=dw0, dw_koff=2 * H)
            # per-node projections: ps = x W0s^T (sender side), pr = x W0r^T + b0 (receiver side)
            ops.segment_sum(g1e, s_rowptr, dps_loc, order=s_order)
            if halo is not None:
                pending = halo.return_rows(dps_loc, async_op=True)   # halo senders' sums go back to their owners
            if not saved["fused"]:
                ops.segment_sum(g1e, rowptr, dpr)
            if halo is not None and pending is not None:
                pending.wait()
            ops.train_linear_bwd(dps, scratch, a=x, w=w_e0, koff=0, res=gx, ga=gx, dw=dw0, dw_koff=0)
            ops.train_linear_bwd(dpr, scratch, a=x, w=w_e0, koff=H, res=gx, ga=gx, dw=dw0, dw_koff=H, db=db0)
            grads[pre + "edge_mlp.0.weight"], grads[pre + "edge_mlp.0.bias"] = dw0, db0

        return ge, g2e, g1e, g2n, g1n

    gec = None
    if coarse is not None:
        cg, nc, nc_loc, ec, c_first = coarse["cg"], coarse["nc"], coarse["nc_loc"], coarse["ec"], coarse["first"]
        gxc = _empty(nc, dev)
        ops.gather_rows(gmap, c_first, gxc)           # rows of the mapped-back tensor that came from the coarse nodes
        gec, g2ec, g1ec, _, _ = blocks_backward(list(model.coarse_processors), "coarse_processors", coarse["layers"], cg.topology,
                                                coarse["halo"], gxc, nc, nc_loc, ec)
    ge, g2e, g1e, g2n, g1n = blocks_backward(list(model.processors), "processors", saved["layers"], topo, halo, gx, n, n_loc, e)
    if coarse is not None:
        gx.index_add_(0, c_first, gxc)                # the coarse nodes' latents are rows of the node encoder's output (:182)

    # ---- encoders: y = LN(W1 ReLU(W0 feat + b0) + b1), no gradient wanted for the features
    def encoder_bwd(seq, name, saved_enc, gy, g_a, g_b):
        inp, h1, v = saved_enc
        dg, dbt = vec(), vec()
        ops.train_ln_bwd(v, seq[3].weight.detach(), g_a, dg, dbt, scratch, g=gy, relu_mask=False)
        grads[name + ".3.weight"], grads[name + ".3.bias"] = dg, dbt
        dw, db = g_like(seq[2].weight), vec()
        ops.train_linear_bwd(g_a, scratch, a=h1, w=seq[2].weight.detach(), mask=h1, ga=g_b, dw=dw, db=db)
        grads[name + ".2.weight"], grads[name + ".2.bias"] = dw, db
        dwp, db = torch.empty(H, H, dtype=torch.float32, device=dev), vec()
        ops.train_linear_bwd(g_b, scratch, a=inp, dw=dwp, db=db)
        width = seq[0].weight.shape[1]
        narrowed.append((name + ".0.weight", dwp, lambda t: t[:, :width].contiguous()))
        grads[name + ".0.weight"] = None
        grads[name + ".0.bias"] = db

    encoder_bwd(model.node_encoder, "node_encoder", saved["enc_n"], gx, g2n, g1n)
    if ge is None:                                # no processor blocks: the edge encoder does not reach the loss
        ge = torch.zeros(e, H, dtype=torch.float32, device=dev)
    if g2e is None:
        g2e = _empty(e, dev)
    encoder_bwd(model.edge_encoder, "edge_encoder", saved["enc_e"], ge, g2e, g1e)
    if coarse is not None:
        if gec is None:
            gec = torch.zeros(ec, H, dtype=torch.float32, device=dev)
        if g2ec is None:
            g2ec = _empty(ec, dev)
        encoder_bwd(model.coarse_edge_encoder, "coarse_edge_encoder", coarse["enc_c"], gec, g2ec, g1ec)
    if flush is not None:
        flush()                                   # every weight gradient of the step in one reduction launch
    for name, padded, cut in narrowed:
        grads[name] = cut(padded)
    if halo is not None:
        # each rank holds the sums over its own rows: one all-reduce over all 136 tensors (4.84 MB)
        flat = torch.cat([g.reshape(-1) for g in grads.values()])
        halo.all_reduce_(flat)
        off = 0
        for k, g in grads.items():
            grads[k] = flat[off: off + g.numel()].view_as(g)
            off += g.numel()
    return grads


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, graph, target_norm, *params):
        loss3, saved = forward_saving(model, graph, target_norm)
        ctx.model, ctx.saved_acts = model, saved
        ctx.names = model._param_list()[0]
        return loss3

    @staticmethod
    def backward(ctx, grad3):
        grads = backward(ctx.model, ctx.saved_acts, grad3.to(torch.float32).contiguous())
        ctx.saved_acts = None
        return (None, None, None) + tuple(grads[k] for k in ctx.names)


def loss_components(model, graph, target_norm):
    """[3] tensor (ux, uy, phi) attached to the autograd graph of the model's parameters."""
    return _LossFn.apply(model, graph, target_norm, *model._param_list()[1])


class CapturedTrainStep:
    """One training step of run_train_multiscale.py:137-142 -- ``zero_grad``, ``loss``, ``backward``, ``optimizer.step``
    -- replayed from a captured HIP graph.  For meshes of a few hundred nodes, where the ~320 launches of a step take
    the host longer to issue than the GPU to run (4.4 ms against 2.3 ms at 561 nodes).

        step = CapturedTrainStep(model, FusedAdam(model.parameters(), lr=1e-4, capturable=True))
        for frame in frames:                       # frames of one mesh (same topology)
            total, ux, uy, phi = step(frame)       # detached loss values of this step

    The first call runs eagerly (it creates every buffer), the second is captured and replayed, later ones copy the
    frame into the captured step's input buffers and replay.  What changes from step to step lives in device memory:
    the frame fields, the normalisers' row counts and the optimiser's scalars (``FusedAdam(capturable=True)``; the
    learning rate is read from ``param_groups`` before every replay, so schedulers work).  Same results as the eager
    sequence bit for bit.  A frame with another topology, or a normaliser that stops accumulating, is run eagerly
    (and the step re-captured)."""

    FIELDS = ("world_pos", "phi", "swelling_phi", "node_type", "mat_param", "target")

    def __init__(self, model, optimizer):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("CapturedTrainStep needs FusedAdam(..., capturable=True)")
        if model.__dict__.get("halo") is not None:
            raise NotImplementedError("CapturedTrainStep: not with a partitioned model")
        if model.sample_ratio > 0 and getattr(model, "coarse_start", None) is None:
            raise ValueError("CapturedTrainStep: a multiscale model draws its coarse graph per call (model.coarse_start is "
                             "None); set model.coarse_start to a node id so that the captured step keeps one coarse graph")
        self.model, self.opt = model, optimizer
        self.norms = (model.node_features_normalizer, model.edge_features_normalizer, model.output_normalizer)
        if model.sample_ratio > 0:
            self.norms += (model.coarse_edge_features_normalizer,)
        self.graph = self.static = self.out = None
        self.done = None

    def _eager(self, frame):
        self.opt.zero_grad()
        total, ux, uy, phi = self.model.loss(frame)
        total.backward()
        self.opt.step()
        return total.detach(), ux.detach(), uy.detach(), phi.detach()

    def _accumulating(self):
        return [nr._num_accumulations < nr._max_accumulations for nr in self.norms]

    def _same_mesh(self, frame):
        """The frame's topology and mesh coordinates are those the captured step was built on: the very tensors
        seen before (same object, not modified since: the step keeps them alive, so an address cannot be reused by
        another mesh), or tensors of equal contents (compared on the device)."""
        st = self.static
        if st is None:
            return False
        for k, (seen, version) in zip(("edge_index", "mesh_pos"), self.mesh_key):
            t = frame[k]
            if t is seen and t._version == version:
                continue
            if t.shape != st[k].shape or not torch.equal(t.to(st[k].device, st[k].dtype), st[k]):
                return False
        return True

    def _make_static(self, frame):
        """Private device copies of the frame: the input buffers of the captured step.  The first (eager) step already
        runs on them, so that everything keyed on the graph object (topology, workspaces) exists before the capture."""
        from .graph import MeshGraph
        dev = torch.device(self.model.device)
        self.static = MeshGraph(**{k: v for k, v in frame.items()}).to(dev)
        for k in self.FIELDS:
            if k in self.static:
                self.static[k] = self.static[k].to(torch.float32).clone()
        # the caller's tensors themselves (kept alive) with their version counters
        self.mesh_key = tuple((frame[k], frame[k]._version) for k in ("edge_index", "mesh_pos"))

    def _load(self, frame):
        for k in self.FIELDS:
            if k in self.static and k in frame:
                self.static[k].copy_(frame[k], non_blocking=True)

    def _capture(self):
        dev = torch.device(self.model.device)
        for nr in self.norms:
            nr.use_device_count(True)
        self.flags = self._accumulating()
        counts = [nr._acc_count for nr in self.norms]
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.opt.zero_grad()
            total, ux, uy, phi = self.model.loss(self.static, check_range=False)
            total.backward()
            self.opt.step()
            self.out = (total.detach(), ux.detach(), uy.detach(), phi.detach())
        # the capture ran the host bookkeeping of one step (optimiser step counts, normaliser counts) without
        # executing anything: that bookkeeping belongs to the first replay
        self.rows_added = [nr._acc_count - c for nr, c in zip(self.norms, counts)]
        self.done = torch.cuda.Event()

    def _replay(self):
        self.graph.replay()
        self.done.record()
        return tuple(t.clone() for t in self.out)

    def close(self):
        for nr in self.norms:
            nr.use_device_count(False)
        had_graph = self.graph is not None
        self.graph = self.static = self.out = None
        if had_graph:
            ops.raise_if_out_of_range("CapturedTrainStep")

    def __call__(self, frame):
        from . import ops
        if self.static is None or not self._same_mesh(frame) or (self.graph is not None and self.flags != self._accumulating()):
            if self.graph is not None:
                self.done.synchronize()
                self.close()
            self._make_static(frame)                # another mesh (or the first call): one eager step on fresh buffers
            return self._eager(self.static)
        if self.graph is None:
            self._load(frame)
            self._capture()                         # host bookkeeping of this step done by the capture itself
            out = self._replay()
        else:
            self.done.synchronize()                 # the previous replay has read its scalars and input buffers
            self._load(frame)
            self.opt.advance_host()
            for nr, rows, acc in zip(self.norms, self.rows_added, self.flags):
                if acc:
                    nr.count_replayed(rows)
            out = self._replay()
        ops.poll_out_of_range("CapturedTrainStep")     # non-blocking; the flag is sticky, close() reads it once more
        return out
