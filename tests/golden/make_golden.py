#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (it imports the
reference checkout read-only from /root/reference); the fixtures it writes are plain
input/output data and are what travels to the GPU box.

For every fixture the script first checks that the CPU oracle (oracle/mgn_oracle.py)
reproduces the imported reference bit for bit, then stores the reference's outputs.

Stand-ins for wheels the reference imports but the image lacks:
  torch_scatter.scatter_add      zeros(dim_size, H).scatter_add_(0, index, src)  (torch-scatter 2.1.2)
  torch_geometric.utils.to_undirected   cat reversed edges, sort by row*N+col, unique  (PyG 2.6.1)
  torch_geometric.nn             names only (never executed on the shipped configs)
  torch_geometric.data.Data      a dict with attribute access (the reference only constructs it and reads fields)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

import oracle as O  # noqa: E402


class Frame(dict):
    """Attribute + item access container standing in for a PyG Data frame."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def to(self, device):
        return self

    def __deepcopy__(self, memo):
        return Frame({k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.items()})


def _install_shims():
    ts = types.ModuleType("torch_scatter")

    def scatter_add(src, index, dim=0, out=None, dim_size=None):
        n = dim_size if dim_size is not None else int(index.max()) + 1
        o = src.new_zeros([n, *src.shape[1:]])
        return o.scatter_add_(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src)

    ts.scatter_add, ts.scatter_mean, ts.scatter_max = scatter_add, None, None
    tg = types.ModuleType("torch_geometric")
    tgn = types.ModuleType("torch_geometric.nn")
    tgu = types.ModuleType("torch_geometric.utils")
    tgd = types.ModuleType("torch_geometric.data")

    def to_undirected(edge_index):
        row, col = edge_index[0], edge_index[1]
        row, col = torch.cat([row, col]), torch.cat([col, row])
        n = int(max(row.max(), col.max())) + 1
        key = torch.unique(row * n + col)
        return torch.stack([torch.div(key, n, rounding_mode="floor"), key % n])

    # torch_cluster's knn_graph is absent: the oracle's restatement stands in (only the multiscale fixture calls it)
    tgn.MessagePassing, tgn.knn_graph = object, (lambda x, k, loop=False: O.knn_graph(x, k))
    tgu.softmax, tgu.to_undirected = None, to_undirected
    tgd.Data = Frame               # core/datasetclass.py builds its frames with Data(**fields)
    tg.nn, tg.utils, tg.data = tgn, tgu, tgd
    sys.modules.update({"torch_scatter": ts, "torch_geometric": tg, "torch_geometric.nn": tgn,
                        "torch_geometric.utils": tgu, "torch_geometric.data": tgd})


_install_shims()
sys.path.insert(0, REF)
from core.meshgraphnet import EncodeProcessDecodeMultiScale, EdgeNodeMessagePassing  # noqa: E402
from core.normalization import Normalizer as RefNormalizer  # noqa: E402
from core.datasetclass import cells_to_edge_index as ref_cells_to_edge_index  # noqa: E402
from core.datasetclass import HydrogelDataset as RefHydrogelDataset  # noqa: E402
from core.rollout import rollout as ref_rollout  # noqa: E402


def bit_equal(a, b):
    return a.shape == b.shape and torch.equal(a, b)


def load_ref_norm_stats(path):
    """Reads a reference normaliser pickle without executing anything from it."""
    class Stub(torch.nn.Module):
        pass
    Stub.__module__, Stub.__name__, Stub.__qualname__ = "core.normalization", "Normalizer", "Normalizer"
    with torch.serialization.safe_globals([(Stub, "core.normalization.Normalizer")]):
        o = torch.load(path, map_location="cpu", weights_only=True)
    return (o._acc_sum.clone(), o._acc_sum_squared.clone(), int(o._acc_count), int(o._num_accumulations))


def ref_norm_from(stats, batch, feat, name):
    n = RefNormalizer(batch, feat, name, "cpu")
    n._acc_sum, n._acc_sum_squared = stats[0].clone(), stats[1].clone()
    n._acc_count, n._num_accumulations = stats[2], stats[3]
    return n


def oracle_norm_from(stats):
    return O.NormStats(stats[0].clone(), stats[1].clone(), stats[2], stats[3])


def build_ref_model(cfg, seed):
    torch.manual_seed(seed)
    m = EncodeProcessDecodeMultiScale(cfg.node_in_dim, cfg.edge_in_dim, hidden_size=cfg.hidden_size,
                                      process_steps=cfg.process_steps, coarse_process_steps=0,
                                      node_out_dim=cfg.node_out_dim, attention=False,
                                      with_mat_params=cfg.with_mat_params, sample_ratio=0.0,
                                      time_dim=cfg.time_dim, device="cpu")
    sd = O.init_state_dict(cfg, seed)
    rsd = m.state_dict()
    assert list(rsd.keys()) == list(sd.keys()), "state_dict key order differs"
    for k in sd:
        assert bit_equal(rsd[k], sd[k]), k
    return m, sd


def norm_stats_arrays(prefix, n):
    if isinstance(n, O.NormStats):
        s, q, c, a = n.acc_sum, n.acc_sum_squared, n.acc_count, n.num_accumulations
    else:
        s, q, c, a = n._acc_sum, n._acc_sum_squared, n._acc_count, n._num_accumulations
    return {f"{prefix}_sum": s.numpy().copy(), f"{prefix}_sumsq": q.numpy().copy(),
            f"{prefix}_count": np.int64(c), f"{prefix}_nacc": np.int64(a)}


# --------------------------------------------------------------------------- G1
def g1_edges():
    out = {}
    rng = np.random.default_rng(7)
    meshes = {"g2x2": O.grid_mesh(2, 2)[1], "g4x3": O.grid_mesh(4, 3)[1], "g32x16": O.grid_mesh(32, 16, 0.08, 0.01)[1]}
    pos, cells = O.grid_mesh(6, 5)
    p = rng.permutation(pos.shape[0])
    irr = p[cells]                      # relabelled node ids
    irr = irr[rng.permutation(irr.shape[0])]
    flip = rng.random(irr.shape[0]) < 0.5
    irr[flip] = irr[flip][:, ::-1]      # mixed orientation
    meshes["irregular"] = np.ascontiguousarray(irr[:-7])  # drop a few cells: boundary holes
    meshes["two_tris"] = np.array([[0, 1, 2], [2, 1, 3]], dtype=np.int64)
    for name, c in meshes.items():
        ref = ref_cells_to_edge_index(torch.from_numpy(c)).numpy()
        mine = O.cells_to_edge_index(c)
        assert ref.dtype == mine.dtype == np.int64 and np.array_equal(ref, mine), name
        out[f"{name}_cells"] = c
        out[f"{name}_edge_index"] = ref
    # hand-checked: two triangles sharing edge (1,2)
    assert out["two_tris_edge_index"].tolist() == [[0, 0, 1, 1, 1, 2, 2, 2, 3, 3], [1, 2, 0, 2, 3, 0, 1, 3, 1, 2]]
    np.savez_compressed(os.path.join(HERE, "g1_edges.npz"), **out)
    print("g1_edges: ok", {k: v.shape for k, v in out.items() if k.endswith("edge_index")})


# --------------------------------------------------------------------------- G2
def g2_mp_layer():
    cfg = O.OracleConfig(process_steps=2)
    _, sd = build_ref_model(cfg, 11)
    ref_model, _ = build_ref_model(cfg, 11)
    g = O.synthetic_graph(8, 4, seed=3)
    n, ne = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, 128, generator=gen)
    e = torch.randn(ne, 128, generator=gen)
    with torch.no_grad():
        rx, re = ref_model.processors[1](x, g["edge_index"], e)
        ox, oe = O.mp_layer(sd, 1, x, g["edge_index"], e)
    assert bit_equal(rx, ox) and bit_equal(re, oe)
    np.savez_compressed(os.path.join(HERE, "g2_mp_layer.npz"), nx=8, ny=4, graph_seed=3, weight_seed=11,
                        layer=1, process_steps=2, sd_digest=O.state_dict_digest(sd),
                        x=x.numpy(), e=e.numpy(), edge_index=g["edge_index"].numpy(),
                        x_out=rx.numpy(), e_out=re.numpy())
    print("g2_mp_layer: ok", n, ne)


# --------------------------------------------------------------------------- G3/G4/G5 (bending)
BEND = os.path.join(REF, "selected_model/bending_best/best_model")


def bending_setup(seed=0):
    cfg = O.OracleConfig()  # node 10, edge 7, H 128, 10 steps, T 1, no mat params
    ref_model, sd = build_ref_model(cfg, seed)
    stats = {k: load_ref_norm_stats(os.path.join(BEND, f"{f}.pth")) for k, f in
             (("node", "node_features_normalizer"), ("edge", "edge_features_normalizer"), ("output", "output_normalizer"))}
    ref_model.node_features_normalizer = ref_norm_from(stats["node"], 1, 10, "node_features_normalizer")
    ref_model.edge_features_normalizer = ref_norm_from(stats["edge"], 1, 7, "edge_features_normalizer")
    ref_model.output_normalizer = ref_norm_from(stats["output"], 1, 3, "output_normalizer")
    norms = {k: oracle_norm_from(v) for k, v in stats.items()}
    return cfg, ref_model, sd, stats, norms


def g3_predict_bending():
    cfg, ref_model, sd, stats, norms = bending_setup()
    g = O.synthetic_graph(32, 16, seed=0, lx=0.08, ly=0.01)
    out = {"nx": 32, "ny": 16, "lx": 0.08, "ly": 0.01, "graph_seed": 0, "weight_seed": 0,
           "sd_digest": O.state_dict_digest(sd)}
    for k, s in stats.items():
        out.update({f"in_{k}_sum": s[0].numpy(), f"in_{k}_sumsq": s[1].numpy(),
                    f"in_{k}_count": np.int64(s[2]), f"in_{k}_nacc": np.int64(s[3])})
    for k in ("mesh_pos", "world_pos", "phi", "swelling_phi", "node_type", "mat_param"):
        out[f"g_{k}"] = g[k].numpy()
    with torch.no_grad():
        rf = ref_model.forward(g)
        of, lat = O.forward(sd, cfg, norms, g, return_latents=True)
        assert bit_equal(rf, of)
        rp = ref_model.predict(g)
        op = O.predict(sd, cfg, norms, g)
        assert bit_equal(rp, op)
    for k, rn in (("node", ref_model.node_features_normalizer), ("edge", ref_model.edge_features_normalizer),
                  ("output", ref_model.output_normalizer)):
        assert bit_equal(rn._acc_sum, norms[k].acc_sum) and rn._acc_count == norms[k].acc_count
        assert rn._num_accumulations == norms[k].num_accumulations
        out.update(norm_stats_arrays(f"out_{k}", rn))
    out["forward"], out["predict2"] = rf.numpy(), rp.numpy()   # predict2 = result of the 2nd call (stats moved twice)
    out["x_enc_rowsum"] = lat["x_enc"].double().sum(1).numpy()
    out["x_proc_rowsum"] = lat["x_proc"].double().sum(1).numpy()
    np.savez_compressed(os.path.join(HERE, "g3_predict_bending.npz"), **out)
    print("g3_predict_bending: ok", rp.shape, float(rp.abs().max()))


def g3b_small_general():
    """time_dim = 2, with_mat_params, fresh normalisers, small mesh, 3 steps."""
    cfg = O.OracleConfig(node_in_dim=2 + 1 + 3 + 5 + 2, process_steps=3, with_mat_params=True, time_dim=2)
    ref_model, sd = build_ref_model(cfg, 21)
    norms = O.make_norms(cfg)
    g = O.synthetic_graph(8, 4, seed=4, time_dim=2, with_target=True)
    with torch.no_grad():
        rp1 = ref_model.predict(g); op1 = O.predict(sd, cfg, norms, g)
        rp2 = ref_model.predict(g); op2 = O.predict(sd, cfg, norms, g)
    assert bit_equal(rp1, op1) and bit_equal(rp2, op2)
    rl = ref_model.loss(g); ol = O.loss(sd, cfg, norms, g)
    for a, b in zip(rl, ol):
        assert bit_equal(a.detach(), b.detach())
    out = {"nx": 8, "ny": 4, "graph_seed": 4, "weight_seed": 21, "time_dim": 2, "process_steps": 3,
           "sd_digest": O.state_dict_digest(sd), "predict1": rp1.numpy(), "predict2": rp2.numpy(),
           "loss": np.array([float(v.detach()) for v in rl], dtype=np.float64)}
    for k, rn in (("node", ref_model.node_features_normalizer), ("edge", ref_model.edge_features_normalizer),
                  ("output", ref_model.output_normalizer)):
        out.update(norm_stats_arrays(f"out_{k}", rn))
    np.savez_compressed(os.path.join(HERE, "g3b_small_general.npz"), **out)
    print("g3b_small_general: ok")


def g4_loss_grads():
    cfg, ref_model, sd, stats, norms = bending_setup()
    g = O.synthetic_graph(32, 16, seed=1, lx=0.08, ly=0.01, with_target=True)
    ref_model.zero_grad()
    rl = ref_model.loss(g)
    rl[0].backward()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ol = O.loss(sdg, cfg, norms, g)
    ol[0].backward()
    out = {"graph_seed": 1, "weight_seed": 0, "loss": np.array([float(v.detach()) for v in rl], dtype=np.float64)}
    names, l2, asum = [], [], []
    for (k, p) in ref_model.named_parameters():
        # the reference's own backward is not run-to-run reproducible on CPU (threaded
        # accumulation in the index / scatter backward), so gradients pin to a tolerance
        rel = float((p.grad - sdg[k].grad).double().norm() / p.grad.double().norm().clamp_min(1e-30))
        assert rel < 2e-6, (k, rel)
        names.append(k); l2.append(float(p.grad.double().norm())); asum.append(float(p.grad.double().sum()))
        if p.grad.numel() <= 128 * 3:
            out["grad__" + k] = p.grad.numpy().copy()
    for a, b in zip(rl, ol):
        assert bit_equal(a.detach(), b.detach())
    out["grad_names"] = np.array(names); out["grad_l2"] = np.array(l2); out["grad_sum"] = np.array(asum)
    np.savez_compressed(os.path.join(HERE, "g4_loss_grads.npz"), **out)
    print("g4_loss_grads: ok", out["loss"])


def g5_rollout(T=100):
    cfg, ref_model, sd, stats, norms = bending_setup()
    g0 = O.synthetic_graph(32, 16, seed=2, lx=0.08, ly=0.01)
    n = g0["mesh_pos"].shape[0]
    gen = torch.Generator().manual_seed(9)
    # synthetic "ground truth": smooth drift + noise; swelling ramps on the top row
    t = torch.arange(T, dtype=torch.float32).view(T, 1, 1)
    world = g0["world_pos"].unsqueeze(0) + 1e-5 * t * torch.randn(1, n, 2, generator=gen)
    phi = g0["phi"].unsqueeze(0) + 1e-4 * t * torch.rand(1, n, 1, generator=gen)
    swell = (0.35 + 0.002 * t.view(T, 1)) * g0["node_type"][:, 4].view(1, n)
    frames = [Frame(mesh_pos=g0["mesh_pos"], node_type=g0["node_type"], mat_param=g0["mat_param"],
                    cells=g0["cells"], edge_index=g0["edge_index"], time=torch.tensor(float(k)),
                    world_pos=world[k], phi=phi[k], swelling_phi=swell[k]) for k in range(T)]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        rr = ref_rollout(ref_model, frames)
    static = {k: g0[k] for k in ("mesh_pos", "node_type", "mat_param", "edge_index")}
    op, ormse = O.rollout(sd, cfg, norms, static, world, phi, swell)
    assert bit_equal(rr["pred"], op)
    assert bit_equal(torch.stack([rr["rmse_x"], rr["rmse_y"], rr["rmse_phi"]]), ormse)
    out = {"T": T, "graph_seed": 2, "traj_seed": 9, "weight_seed": 0,
           "world_pos": world.numpy(), "phi": phi.numpy(), "swelling": swell.numpy(),
           "pred": rr["pred"].numpy(), "rmse": ormse.numpy()}
    for k, rn in (("node", ref_model.node_features_normalizer), ("edge", ref_model.edge_features_normalizer)):
        out.update(norm_stats_arrays(f"out_{k}", rn))
    np.savez_compressed(os.path.join(HERE, "g5_rollout.npz"), **out)
    print("g5_rollout: ok", rr["pred"].shape, ormse)


def g6_dataset():
    """The reference's own HydrogelDataset (core/datasetclass.py:22-119) on a small synthetic trajectory in the
    schema of gt_gen/gt_gen/bending_gen.py:297-306: frames with and without targets, time_dim 1 and 2, noise off
    and on (torch.manual_seed(123) right before the call; the normal draws of :77 and :84 are stored too, in call
    order, so that the device noise kernel can be checked with the very same draws)."""
    import tempfile
    nx, ny, T = 8, 4, 7
    rng = np.random.default_rng(42)
    pos, cells = O.grid_mesh(nx, ny, lx=0.08, ly=0.01)
    n = pos.shape[0]
    sample = dict(mesh_coords=pos.astype(np.float32), cells=cells, node_type=O.bending_node_type(nx, ny).astype(np.float32),
                  u_time_series=(1e-3 * rng.standard_normal((T, n, 2))).astype(np.float32),
                  swell_time_series=np.linspace(0.30, 0.42, T).astype(np.float32), t=(0.5 * np.arange(T)).astype(np.float32),
                  chi=np.float32(0.2), diffusivity=np.float32(7.5e-6))
    sample["φ_time_series"] = (0.5 + 0.1 * rng.random((T, n))).astype(np.float32)
    out = {"npz_" + k: v for k, v in sample.items()}
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, "traj_a.npz"), **sample)
        for add_targets in (True, False):
            for time_dim in (1, 2):
                for noise in (0.0, 0.02):
                    if not add_targets and noise > 0:
                        continue                      # the reference ignores noise_level without targets (:109-119)
                    ds = RefHydrogelDataset(d, noise_level=noise, add_targets=add_targets, time_dim=time_dim)
                    assert len(ds) == 1 and ds.get_name(0) == "traj_a"
                    torch.manual_seed(123)
                    frames = ds[0]
                    tag = f"tgt{int(add_targets)}_td{time_dim}_noise{int(noise > 0)}"
                    out[tag + "_n_frames"] = len(frames)
                    for k in ("time", "world_pos", "phi", "swelling_phi") + (("target",) if add_targets else ()):
                        out[f"{tag}_{k}"] = torch.stack([torch.as_tensor(fr[k]) for fr in frames]).numpy()
                    fr0 = frames[0]
                    for k in ("mesh_pos", "node_type", "mat_param", "cells", "edge_index"):
                        assert all(fr[k] is fr0[k] for fr in frames)
                    if tag == "tgt1_td1_noise0":
                        for k in ("mesh_pos", "node_type", "mat_param", "cells", "edge_index"):
                            out["static_" + k] = fr0[k].numpy()
                    if noise > 0:
                        torch.manual_seed(123)        # the same draws, in the order of :77 and :84 per frame
                        zw, zp = [], []
                        for fr in frames:
                            zw.append(torch.randn(n, 2)); zp.append(torch.randn(n, 1))
                        out[tag + "_z_world"], out[tag + "_z_phi"] = torch.stack(zw).numpy(), torch.stack(zp).numpy()
                        out[tag + "_noise_level"] = np.float32(noise)
    np.savez_compressed(os.path.join(HERE, "g6_dataset.npz"), **out)
    print("g6 dataset frames ok:", sorted(k for k in out if k.endswith("_n_frames")))


def g7_attention_multiscale():
    """The two options no shipped config enables, through the reference's own forward / predict
    (core/meshgraphnet.py:100-108, :179-194, :237-283): attention alone, the coarse branch alone, and both.
    The random first index of the farthest-point sampling (:248) is fixed by torch.manual_seed right before each
    call and recorded; knn_graph is the oracle's restatement on both sides (see _install_shims)."""
    g = O.synthetic_graph(12, 8, seed=4)
    n = g["mesh_pos"].shape[0]
    out = {"nx": 12, "ny": 8, "graph_seed": 4}
    for tag, att, ratio, csteps in (("att", True, 0.0, 0), ("ms", False, 0.5, 2), ("both", True, 0.4, 2)):
        cfg = O.OracleConfig(process_steps=3, attention=att, sample_ratio=ratio, coarse_process_steps=csteps)
        torch.manual_seed(21)
        m = EncodeProcessDecodeMultiScale(cfg.node_in_dim, cfg.edge_in_dim, hidden_size=128, process_steps=3,
                                          coarse_process_steps=csteps, node_out_dim=3, attention=att,
                                          with_mat_params=False, sample_ratio=ratio, time_dim=1, device="cpu")
        sd = O.init_state_dict(cfg, 21)
        rsd = m.state_dict()
        assert list(rsd.keys()) == list(sd.keys()), (tag, "state_dict key order differs")
        for k in sd:
            assert bit_equal(rsd[k], sd[k]), (tag, k)
        norms = O.make_norms(cfg)
        with torch.no_grad():
            torch.manual_seed(5)
            start1 = int(torch.randint(0, n, (1,)))
            torch.manual_seed(5)
            rf = m.forward(g)
            of = O.forward(sd, cfg, norms, g, coarse_start=start1)
            assert bit_equal(rf, of), tag
            torch.manual_seed(6)
            start2 = int(torch.randint(0, n, (1,)))
            torch.manual_seed(6)
            rp = m.predict(g)
            op = O.predict(sd, cfg, norms, g, coarse_start=start2)
            assert bit_equal(rp, op), tag
        out.update({f"{tag}_forward": rf.numpy(), f"{tag}_predict2": rp.numpy(), f"{tag}_start1": start1, f"{tag}_start2": start2,
                    f"{tag}_sd_digest": O.state_dict_digest(sd), f"{tag}_n_keys": len(sd)})
        if ratio > 0:
            ce, cei, first = O.coarse_graph(cfg, g, start1)
            out.update({f"{tag}_first1": first.numpy(), f"{tag}_coarse_edge_index1": cei.numpy()})
            rn = m.coarse_edge_features_normalizer
            assert bit_equal(rn._acc_sum, norms["coarse_edge"].acc_sum) and rn._acc_count == norms["coarse_edge"].acc_count
            out.update(norm_stats_arrays(f"{tag}_out_coarse_edge", rn))
    np.savez_compressed(os.path.join(HERE, "g7_attention_multiscale.npz"), **out)
    print("g7_attention_multiscale: ok")


def g8_options_loss_grads():
    """``loss`` + ``backward`` of the reference itself (core/meshgraphnet.py:285-304 through torch autograd) with the
    attention option, the coarse branch, and both: loss values, and per parameter the norm and the sum of its gradient
    (the small tensors -- biases, LayerNorm vectors -- in full).  Pins the backward of the two options, including what
    autograd makes of the in-place clamp through the repeated receiver index (:106).  The oracle's autograd must
    agree to 2e-6 (the reference's backward is not run-to-run reproducible on CPU)."""
    g = O.synthetic_graph(14, 9, seed=6, with_target=True)
    n = g["mesh_pos"].shape[0]
    out = {"nx": 14, "ny": 9, "graph_seed": 6, "weight_seed": 23}
    for tag, att, ratio, csteps in (("att", True, 0.0, 0), ("ms", False, 0.5, 2), ("both", True, 0.4, 1)):
        cfg = O.OracleConfig(process_steps=2, attention=att, sample_ratio=ratio, coarse_process_steps=csteps)
        m = EncodeProcessDecodeMultiScale(cfg.node_in_dim, cfg.edge_in_dim, hidden_size=128, process_steps=2,
                                          coarse_process_steps=csteps, node_out_dim=3, attention=att,
                                          with_mat_params=False, sample_ratio=ratio, time_dim=1, device="cpu")
        sd = O.init_state_dict(cfg, 23)
        m.load_state_dict(sd)
        norms = O.make_norms(cfg)
        torch.manual_seed(8)
        start = int(torch.randint(0, n, (1,)))
        torch.manual_seed(8)
        m.zero_grad()
        rl = m.loss(g)
        rl[0].backward()
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ol = O.loss(sdg, cfg, norms, g, coarse_start=start)
        ol[0].backward()
        for a, b in zip(rl, ol):
            assert bit_equal(a.detach(), b.detach()), tag
        names, l2, asum = [], [], []
        for k, p in m.named_parameters():
            assert (p.grad is None) == (sdg[k].grad is None), (tag, k)
            if p.grad is None:
                continue
            rel = float((p.grad - sdg[k].grad).double().norm() / p.grad.double().norm().clamp_min(1e-30))
            assert rel < 2e-6, (tag, k, rel)
            names.append(k); l2.append(float(p.grad.double().norm())); asum.append(float(p.grad.double().sum()))
            if p.grad.numel() <= 128:
                out[f"{tag}_grad__{k}"] = p.grad.numpy().copy()
        out.update({f"{tag}_loss": np.array([float(v.detach()) for v in rl], dtype=np.float64), f"{tag}_start": start,
                    f"{tag}_grad_names": np.array(names), f"{tag}_grad_l2": np.array(l2), f"{tag}_grad_sum": np.array(asum),
                    f"{tag}_sd_digest": O.state_dict_digest(sd)})
    np.savez_compressed(os.path.join(HERE, "g8_options_loss_grads.npz"), **out)
    print("g8_options_loss_grads: ok", {t: out[f"{t}_loss"][0] for t in ("att", "ms", "both")})


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                       # regenerate only the named fixtures, e.g. `make_golden.py g6_dataset`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_edges()
    g2_mp_layer()
    g3_predict_bending()
    g3b_small_general()
    g4_loss_grads()
    g5_rollout()
    g6_dataset()
    g7_attention_multiscale()
    g8_options_loss_grads()
    print("all fixtures written to", HERE)
