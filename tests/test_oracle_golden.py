"""CPU: the oracle against the golden vectors written from the imported reference.
Integer results must be identical; float results are compared at 2e-6 (the fixtures were
bit-identical in the process that wrote them, but another host may pick another CPU
kernel / thread split inside torch)."""
import pytest
import numpy as np
import torch

import oracle as O
from conftest import GOLDEN, rel_l2

FT = 2e-6


def test_edge_build_golden():
    d = np.load(f"{GOLDEN}/g1_edges.npz")
    for name in ("g2x2", "g4x3", "g32x16", "irregular", "two_tris"):
        ei = O.cells_to_edge_index(d[f"{name}_cells"])
        assert ei.dtype == np.int64 and np.array_equal(ei, d[f"{name}_edge_index"]), name
    assert d["g32x16_edge_index"].shape == (2, 3168)             # bending mesh: 561 nodes, 3168 edges
    assert O.cells_to_edge_index(np.zeros((0, 3), dtype=np.int64)).shape == (2, 0)


def test_edge_build_sorted_unique_symmetric():
    _, cells = O.grid_mesh(9, 7)
    ei = O.cells_to_edge_index(cells)
    key = ei[0] * 80 + ei[1]
    assert np.all(np.diff(key) > 0)
    assert set(map(tuple, ei.T)) == set(map(tuple, ei[::-1].T))


def test_receiver_csr_properties():
    _, cells = O.grid_mesh(6, 4)
    ei = O.cells_to_edge_index(cells)
    rowptr, snd, rcv, perm = O.receiver_csr(ei, 35)
    assert rowptr[0] == 0 and rowptr[-1] == ei.shape[1] and np.all(np.diff(rcv) >= 0)
    assert np.array_equal(ei[0][perm], snd) and np.array_equal(ei[1][perm], rcv)
    for n in range(35):
        assert np.all(rcv[rowptr[n]:rowptr[n + 1]] == n)
    # symmetric graph sorted by (sender, receiver): receiver-major order is the row swap
    assert np.array_equal(snd, ei[1]) and np.array_equal(rcv, ei[0])


def test_mp_layer_golden():
    d = np.load(f"{GOLDEN}/g2_mp_layer.npz")
    sd = O.init_state_dict(O.OracleConfig(process_steps=int(d["process_steps"])), int(d["weight_seed"]))
    assert abs(O.state_dict_digest(sd) - float(d["sd_digest"])) < 1e-9 * abs(float(d["sd_digest"]))
    with torch.no_grad():
        x, e = O.mp_layer(sd, int(d["layer"]), torch.from_numpy(d["x"]), torch.from_numpy(d["edge_index"]), torch.from_numpy(d["e"]))
    assert rel_l2(x, torch.from_numpy(d["x_out"])) < FT and rel_l2(e, torch.from_numpy(d["e_out"])) < FT


def test_projected_formulation_matches():
    """The per-node projection form of the first edge Linear equals the concatenated form."""
    d = np.load(f"{GOLDEN}/g2_mp_layer.npz")
    sd = O.init_state_dict(O.OracleConfig(process_steps=2), int(d["weight_seed"]))
    ei = d["edge_index"]
    rowptr, snd, rcv, perm = O.receiver_csr(ei, d["x"].shape[0])
    x2, e2 = O.mp_layer_csr_projected(sd, 1, torch.from_numpy(d["x"]), rowptr, snd, rcv, torch.from_numpy(d["e"][perm]))
    assert rel_l2(x2, torch.from_numpy(d["x_out"])) < FT
    assert rel_l2(e2, torch.from_numpy(d["e_out"][perm])) < FT


def _bending():
    d = np.load(f"{GOLDEN}/g3_predict_bending.npz")
    cfg = O.OracleConfig()
    sd = O.init_state_dict(cfg, int(d["weight_seed"]))
    norms = {k: O.NormStats(torch.from_numpy(d[f"in_{k}_sum"]).clone(), torch.from_numpy(d[f"in_{k}_sumsq"]).clone(),
                            int(d[f"in_{k}_count"]), int(d[f"in_{k}_nacc"])) for k in ("node", "edge", "output")}
    return d, cfg, sd, norms


def test_bending_normalizer_facts():
    d, cfg, sd, norms = _bending()
    # 561 nodes / 3168 edges per accumulation, 195111 accumulations (epoch 489 of the shipped run)
    assert norms["node"].acc_count == 561 * norms["node"].num_accumulations == 109457271
    assert norms["edge"].acc_count == 3168 * norms["edge"].num_accumulations
    assert norms["node"].num_accumulations == 195111 < norms["node"].max_accumulations
    assert len(sd) == 136 and sum(v.numel() for v in sd.values()) == 1209987


def test_predict_bending_golden():
    d, cfg, sd, norms = _bending()
    g = O.synthetic_graph(32, 16, seed=int(d["graph_seed"]), lx=float(d["lx"]), ly=float(d["ly"]))
    for k in ("mesh_pos", "world_pos", "phi", "swelling_phi", "node_type"):
        assert np.array_equal(g[k].numpy(), d[f"g_{k}"]), k
    with torch.no_grad():
        f = O.forward(sd, cfg, norms, g)
        p = O.predict(sd, cfg, norms, g)
    assert rel_l2(f, torch.from_numpy(d["forward"])) < FT and rel_l2(p, torch.from_numpy(d["predict2"])) < FT
    for k in ("node", "edge", "output"):            # normalisers keep accumulating in predict
        assert norms[k].acc_count == int(d[f"out_{k}_count"]) and norms[k].num_accumulations == int(d[f"out_{k}_nacc"])
        assert rel_l2(norms[k].acc_sum, torch.from_numpy(d[f"out_{k}_sum"])) < 1e-7
    assert norms["node"].num_accumulations == int(d["in_node_nacc"]) + 2
    assert norms["output"].num_accumulations == int(d["in_output_nacc"])


def test_general_config_golden():
    d = np.load(f"{GOLDEN}/g3b_small_general.npz")
    cfg = O.OracleConfig(node_in_dim=13, process_steps=int(d["process_steps"]), with_mat_params=True, time_dim=2)
    sd = O.init_state_dict(cfg, int(d["weight_seed"]))
    norms = O.make_norms(cfg)
    g = O.synthetic_graph(int(d["nx"]), int(d["ny"]), seed=int(d["graph_seed"]), time_dim=2, with_target=True)
    with torch.no_grad():
        p1 = O.predict(sd, cfg, norms, g); p2 = O.predict(sd, cfg, norms, g)
        ls = O.loss(sd, cfg, norms, g)
    assert rel_l2(p1, torch.from_numpy(d["predict1"])) < FT and rel_l2(p2, torch.from_numpy(d["predict2"])) < FT
    assert np.allclose([float(v) for v in ls], d["loss"], rtol=1e-5)


def test_loss_grads_golden():
    d4 = np.load(f"{GOLDEN}/g4_loss_grads.npz")
    d, cfg, sd, norms = _bending()
    g = O.synthetic_graph(32, 16, seed=int(d4["graph_seed"]), lx=0.08, ly=0.01, with_target=True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ls = O.loss(sdg, cfg, norms, g)
    ls[0].backward()
    assert np.allclose([float(v.detach()) for v in ls], d4["loss"], rtol=1e-5)
    for name, l2 in zip(d4["grad_names"], d4["grad_l2"]):
        got = float(sdg[str(name)].grad.double().norm())
        assert abs(got - l2) <= 1e-4 * max(l2, 1e-12), name
    for k in d4.files:
        if k.startswith("grad__"):
            assert rel_l2(sdg[k[6:]].grad, torch.from_numpy(d4[k])) < 1e-4, k


def test_rollout_golden():
    d5 = np.load(f"{GOLDEN}/g5_rollout.npz")
    d, cfg, sd, norms = _bending()
    g0 = O.synthetic_graph(32, 16, seed=int(d5["graph_seed"]), lx=0.08, ly=0.01)
    static = {k: g0[k] for k in ("mesh_pos", "node_type", "mat_param", "edge_index")}
    T = 20      # the first 20 of the 100 stored steps keep this test short
    pred, rmse = O.rollout(sd, cfg, norms, static, torch.from_numpy(d5["world_pos"][:T]), torch.from_numpy(d5["phi"][:T]),
                           torch.from_numpy(d5["swelling"][:T]))
    ref = torch.from_numpy(d5["pred"][:T])
    assert pred.shape == ref.shape and rel_l2(pred, ref) < 1e-5
    assert torch.equal(pred[0], torch.cat([torch.from_numpy(d5["world_pos"][0]), torch.from_numpy(d5["phi"][0])], -1))


@pytest.mark.parametrize("tag,att,ratio,csteps", [("att", True, 0.0, 0), ("ms", False, 0.5, 2), ("both", True, 0.4, 2)])
def test_oracle_attention_and_multiscale_golden(tag, att, ratio, csteps):
    """Fixture G7: forward / predict of the reference model with the attention option, the coarse multiscale branch,
    and both (core/meshgraphnet.py:100-108, :179-194, :237-283), bit for bit."""
    d = np.load(f"{GOLDEN}/g7_attention_multiscale.npz")
    cfg = O.OracleConfig(process_steps=3, attention=att, sample_ratio=ratio, coarse_process_steps=csteps)
    sd = O.init_state_dict(cfg, 21)
    assert len(sd) == int(d[f"{tag}_n_keys"])
    assert abs(O.state_dict_digest(sd) - float(d[f"{tag}_sd_digest"])) < 1e-9 * abs(float(d[f"{tag}_sd_digest"]))
    g = O.synthetic_graph(int(d["nx"]), int(d["ny"]), seed=int(d["graph_seed"]))
    norms = O.make_norms(cfg)
    with torch.no_grad():
        f = O.forward(sd, cfg, norms, g, coarse_start=int(d[f"{tag}_start1"]))
        p = O.predict(sd, cfg, norms, g, coarse_start=int(d[f"{tag}_start2"]))
    assert torch.equal(f, torch.from_numpy(d[f"{tag}_forward"])) and torch.equal(p, torch.from_numpy(d[f"{tag}_predict2"]))
    if ratio > 0:
        ce, cei, first = O.coarse_graph(cfg, g, int(d[f"{tag}_start1"]))
        assert np.array_equal(first.numpy(), d[f"{tag}_first1"]) and np.array_equal(cei.numpy(), d[f"{tag}_coarse_edge_index1"])
        assert norms["coarse_edge"].acc_count == int(d[f"{tag}_out_coarse_edge_count"])


@pytest.mark.parametrize("tag,att,ratio,csteps", [("att", True, 0.0, 0), ("ms", False, 0.5, 2), ("both", True, 0.4, 1)])
def test_oracle_options_loss_grads_golden(tag, att, ratio, csteps):
    """Fixture G8: ``loss`` + ``backward`` of the reference model itself with attention, the coarse branch and both
    (fresh statistics, first sample of the coarse graph as drawn by the reference): the oracle's loss values, the
    norm of every parameter gradient and the small gradient tensors in full.  This is what pins the backward of the
    attention weights, including the in-degree factor autograd produces for ``alpha[col] = alpha[col].clamp(...)``
    (core/meshgraphnet.py:106)."""
    d = np.load(f"{GOLDEN}/g8_options_loss_grads.npz")
    cfg = O.OracleConfig(process_steps=2, attention=att, sample_ratio=ratio, coarse_process_steps=csteps)
    sd = O.init_state_dict(cfg, int(d["weight_seed"]))
    assert O.state_dict_digest(sd) == float(d[f"{tag}_sd_digest"])
    g = O.synthetic_graph(int(d["nx"]), int(d["ny"]), seed=int(d["graph_seed"]), with_target=True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ls = O.loss(sdg, cfg, O.make_norms(cfg), g, coarse_start=int(d[f"{tag}_start"]))
    ls[0].backward()
    assert np.allclose([float(v.detach()) for v in ls], d[f"{tag}_loss"], rtol=1e-6)
    names = [str(k) for k in d[f"{tag}_grad_names"]]
    assert any("attn_lin" in k for k in names) == att and any(k.startswith("coarse_") for k in names) == (ratio > 0)
    for name, l2 in zip(names, d[f"{tag}_grad_l2"]):
        got = float(sdg[name].grad.double().norm())
        assert abs(got - l2) <= 1e-4 * max(l2, 1e-12), name
    pre = f"{tag}_grad__"
    for k in d.files:
        if k.startswith(pre):
            assert rel_l2(sdg[k[len(pre):]].grad, torch.from_numpy(d[k])) < 1e-4, k


def test_attention_gradient_carries_the_in_degree_factor():
    """The reference's expression for the message weights (core/meshgraphnet.py:101-107) differentiated by torch, against
    the closed form the HIP backward uses: g_alpha = gw / D - deg * [S >= 1e-6] * sum(gw * alpha) / D^2."""
    gen = torch.Generator().manual_seed(0)
    rcv = torch.tensor([0, 0, 0, 1, 1, 2, 4, 4, 4, 4])
    al = torch.rand(10, dtype=torch.float64, generator=gen).requires_grad_(True)
    den = torch.zeros(5, dtype=torch.float64).scatter_add_(0, rcv, al)
    den[rcv] = den[rcv].clamp(min=1e-6)
    gw = torch.rand(10, dtype=torch.float64, generator=gen)
    ((al / den[rcv]) * gw).sum().backward()
    a = al.detach()
    S = torch.zeros(5, dtype=torch.float64).scatter_add_(0, rcv, a)
    t = torch.zeros(5, dtype=torch.float64).scatter_add_(0, rcv, gw * a)
    deg = torch.bincount(rcv, minlength=5).double()
    closed = gw / S[rcv] - deg[rcv] * t[rcv] / S[rcv] ** 2
    assert torch.allclose(al.grad, closed, rtol=1e-12, atol=1e-14)
