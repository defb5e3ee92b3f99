"""GPU parity of the training step (loss + gradients of all parameters) against the golden
vectors generated from the reference's own autograd and against the CPU oracle's autograd.

Tolerance: loss values rel 1e-5.  Gradients are checked twice:

 * tightly (rel-L2 <= 1e-5 per parameter tensor; measured 3e-7 .. 3e-6) against the oracle's autograd in fp64 with the
   ReLU decisions pinned to the ones the GPU forward took.  The gradient of a ReLU network is
   discontinuous -- a hidden pre-activation within rounding of zero flips its mask and every
   layer below inherits the jump -- so two correct fp32 evaluations of the same network agree
   only up to those flips; with the pattern pinned the comparison is exact arithmetic parity;
 * loosely against the fp32 reference gradients themselves (golden fixture / oracle autograd),
   bound 2e-3.  For scale: on the bending fixture the reference's own fp32 gradients differ
   from an fp64 evaluation of the same function by 4e-5 (processor block 5) up to 9e-4 (node
   encoder), and the fp32 oracle run on two hosts differs by 4e-5."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import GOLDEN, rel_l2
from test_gpu_model import DEV, bending_fixture, build_model, copy_norms_frozen, graph_to, load_bending_norms

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-5


def test_training_blocks_vs_torch():
    """Each training block against the same operation written in torch fp32."""
    from gnn_amd import ops
    gen = torch.Generator().manual_seed(0)
    rows, n = 1000, 300
    A, B, G = (torch.randn(rows, 128, generator=gen) for _ in range(3))
    W = torch.randn(128, 384, generator=gen) / 16
    bias, gamma, beta = (torch.randn(128, generator=gen) for _ in range(3))
    P, Q = torch.randn(n, 128, generator=gen), torch.randn(n, 128, generator=gen)
    ip, iq = torch.randint(0, n, (rows,), generator=gen, dtype=torch.int32), torch.randint(0, n, (rows,), generator=gen, dtype=torch.int32)
    d = lambda t: t.to(DEV)
    scratch = torch.empty(ops.train_scratch_bytes(), dtype=torch.uint8, device=DEV)
    out = torch.empty(rows, 128, device=DEV)
    # linear forward: two blocks + bias + gathered addends + ReLU
    ops.train_linear_fwd(d(A), d(W), 128, out, b=d(B), koff_b=256, bias=d(bias), p=d(P), idx_p=d(ip), q=d(Q), idx_q=d(iq), relu=True)
    ref = torch.relu(A @ W[:, 128:256].T + B @ W[:, 256:].T + bias + P[ip.long()] + Q[iq.long()])
    assert rel_l2(out, ref) < 2e-6, rel_l2(out, ref)
    ops.train_linear_fwd(d(A), d(W), 0, out)
    assert rel_l2(out, A @ W[:, :128].T) < 2e-6
    # layer norm forward / backward
    v = torch.relu(A).requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(v, (128,), gm, bt, 1e-5)
    y, yres = torch.empty_like(out), torch.empty_like(out)
    ops.train_ln_fwd(d(v.detach()), d(gamma), d(beta), y=y, res=d(B), yres=yres)
    assert rel_l2(y, y_ref.detach()) < 2e-6 and rel_l2(yres, y_ref.detach() + B) < 2e-6
    Gn = torch.randn(n, 128, generator=gen)
    up = G + Gn[iq.long()]
    y_ref.backward(up)
    gpre, dg, dbt = torch.empty_like(out), torch.empty(128, device=DEV), torch.empty(128, device=DEV)
    ops.train_ln_bwd(d(v.detach()), d(gamma), gpre, dg, dbt, scratch, g=d(G), gg=d(Gn), idx=d(iq), relu_mask=True)
    ref_pre = v.grad * (v.detach() > 0)
    assert rel_l2(gpre, ref_pre) < 5e-6, rel_l2(gpre, ref_pre)
    assert rel_l2(dg, gm.grad) < 5e-6 and rel_l2(dbt, bt.grad) < 5e-6
    # linear backward on a column block
    mask = torch.relu(B)
    GA, dW, db = torch.empty_like(out), torch.zeros(128, 384, device=DEV), torch.empty(128, device=DEV)
    ops.train_linear_bwd(d(G), scratch, a=d(A), w=d(W), koff=256, mask=d(mask), res=d(B), ga=GA, dw=dW, dw_koff=256, db=db)
    assert rel_l2(GA, (G @ W[:, 256:]) * (mask > 0) + B) < 2e-6
    assert rel_l2(dW[:, 256:], G.T @ A) < 5e-6 and float(dW[:, :256].abs().max()) == 0.0
    assert rel_l2(db, G.sum(0)) < 5e-6
    # segmented sums, plain and through an index list, with empty segments
    cnt = torch.randint(0, 8, (n,), generator=gen)
    cnt[5] = 0
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    m = int(rowptr[-1])
    src = torch.randn(m, 128, generator=gen)
    order = torch.randperm(m, generator=gen).to(torch.int32)
    seg = torch.repeat_interleave(torch.arange(n), cnt)
    o1, o2 = torch.empty(n, 128, device=DEV), torch.empty(n, 128, device=DEV)
    ops.segment_sum(d(src), d(rowptr), o1)
    ops.segment_sum(d(src), d(rowptr), o2, order=d(order))
    assert rel_l2(o1, torch.zeros(n, 128).index_add_(0, seg, src)) < 2e-6
    assert rel_l2(o2, torch.zeros(n, 128).index_add_(0, seg, src[order.long()])) < 2e-6
    assert float(o1[5].abs().max()) == 0.0


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 65, 32 * 256 * 3 + 17, 32 * 256 * 4 - 5])
def test_linear_backward_row_counts(rows, matmul_mode):
    """Linear backward against fp64 torch for row counts around the 32-row tile and for counts that give a workgroup
    several tiles, odd and even many (the tile loop runs two passes per round over two register sets): input gradient
    alone, weight gradient alone, both, with and without the ReLU mask and the residual."""
    from gnn_amd import ops
    gen = torch.Generator().manual_seed(rows)
    A, G, M, R = (torch.randn(rows, 128, generator=gen) for _ in range(4))
    W = torch.randn(128, 256, generator=gen) / 16
    d = lambda t: t.to(DEV)
    scratch = torch.empty(ops.train_scratch_bytes(), dtype=torch.uint8, device=DEV)
    G64, A64, W64 = G.double(), A.double(), W.double()
    ga_plain, dw_ref, db_ref = G64 @ W64[:, 128:], G64.T @ A64, G64.sum(0)
    ga_full = ga_plain * (M > 0) + R.double()
    tol = 5e-6
    GA, dW, db = torch.empty(rows, 128, device=DEV), torch.zeros(128, 256, device=DEV), torch.empty(128, device=DEV)
    ops.train_linear_bwd(d(G), scratch, a=d(A), w=d(W), koff=128, mask=d(M), res=d(R), ga=GA, dw=dW, dw_koff=128, db=db)
    assert rel_l2(GA.double(), ga_full) < tol and rel_l2(dW[:, 128:].double(), dw_ref) < tol and rel_l2(db.double(), db_ref) < tol
    assert float(dW[:, :128].abs().max()) == 0.0
    GA2 = torch.empty_like(GA)
    ops.train_linear_bwd(d(G), scratch, w=d(W), koff=128, ga=GA2)                      # input gradient alone, no mask / residual
    assert rel_l2(GA2.double(), ga_plain) < tol
    ops.train_linear_bwd(d(G), scratch, w=d(W), koff=128, res=d(R), ga=GA2)            # residual alone
    assert rel_l2(GA2.double(), ga_plain + R.double()) < tol
    dW2, db2 = torch.zeros(128, 256, device=DEV), torch.empty(128, device=DEV)
    ops.train_linear_bwd(d(G), scratch, a=d(A), dw=dW2, dw_koff=0, db=db2)             # weight gradient alone
    assert rel_l2(dW2[:, :128].double(), dw_ref) < tol and rel_l2(db2.double(), db_ref) < tol
    if matmul_mode != "f32":                                                            # one kernel: same bits either way
        assert torch.equal(dW2[:, :128], dW[:, 128:])


class FixedStats(O.NormStats):
    """Normaliser with the mean / std of `src` held as given (no accumulation), in `dtype`."""

    def __init__(self, src: O.NormStats, dtype):
        super().__init__(src.acc_sum.clone(), src.acc_sum_squared.clone(), src.acc_count, src.num_accumulations, 0)
        self._m, self._s = src.mean().to(dtype), src.std().to(dtype)

    def mean(self):
        return self._m

    def std(self):
        return self._s


def oracle_grads(cfg, sd, norms_after, g, dtype):
    """Oracle autograd in `dtype` with the statistics the fp32 run ended with."""
    sdg = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    gg = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in g.items()}
    ol = O.loss(sdg, cfg, {k: FixedStats(v, dtype) for k, v in norms_after.items()}, gg)
    ol[0].backward()
    return [float(v.detach()) for v in ol], {k: v.grad for k, v in sdg.items()}


LOOSE_TOL = 2e-3


def gpu_loss_and_grads(model, g):
    """loss3, gradients and the activation pattern of one training step, taken below the autograd
    wrapper so that the saved activations are visible."""
    from gnn_amd import training
    gg = graph_to(g, DEV)
    cur = torch.cat([gg["world_pos"], gg["phi"]], -1)
    tgt = model.output_normalizer(gg["target"] - cur).contiguous()
    loss3, saved = training.forward_saving(model, gg, tgt, keep_hidden=True)      # the default path + its ReLU pattern
    grads = training.backward(model, saved, torch.ones(3, device=DEV))
    topo = saved["topo"]
    perm = topo.perm.cpu()

    def orig(t):                                   # CSR edge order -> edge_index order
        out = torch.empty_like(t)
        out[perm] = t
        return out

    masks = [saved["enc_n"][1].cpu() > 0, orig(saved["enc_e"][1].cpu() > 0)]
    for (_x, _e, h1e, ve, _a, h1n, vn, _ps, _pr, _att) in saved["layers"]:
        masks += [orig(h1e.cpu() > 0), orig(ve.cpu() > 0), h1n.cpu() > 0, vn.cpu() > 0]
    masks.append(saved["hd"].cpu() > 0)
    return loss3.cpu(), {k: v.cpu() for k, v in grads.items()}, masks


def pinned_oracle_grads(cfg, sd, norms_after, g, masks):
    """fp64 oracle autograd with the statistics the fp32 run ended with and the given ReLU pattern."""
    dt = torch.float64
    sdg = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    gg = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in g.items()}
    with O.forced_relu_masks(masks):
        ol = O.loss(sdg, cfg, {k: FixedStats(v, dt) for k, v in norms_after.items()}, gg)
    ol[0].backward()
    return [float(v.detach()) for v in ol], {k: v.grad for k, v in sdg.items()}


def check_grads(grads, pinned64, ref32):
    worst_t, worst_l = ("", 0.0), ("", 0.0)
    for k in ref32:
        rt, rl = rel_l2(grads[k].double(), pinned64[k]), rel_l2(grads[k], ref32[k])
        assert rt < GRAD_TOL, (k, "pinned pattern", rt)
        assert rl < LOOSE_TOL, (k, "fp32 reference", rl)
        worst_t = max(worst_t, (k, rt), key=lambda x: x[1])
        worst_l = max(worst_l, (k, rl), key=lambda x: x[1])
    print("worst gradient rel-L2: pinned pattern %s %.3e; fp32 reference %s %.3e" % (worst_t + worst_l))


def run_loss_backward(model, gg):
    model.zero_grad()
    ls = model.loss(gg)
    ls[0].backward()
    return [float(v.detach()) for v in ls], {k: p.grad.detach().cpu() for k, p in model.named_parameters()}


def test_loss_grads_bending_golden():
    """Config 4 parity on the bending mesh: loss(graph) and loss.backward() of the reference
    (real bending_best normaliser statistics, seeded weights)."""
    d4 = np.load(f"{GOLDEN}/g4_loss_grads.npz")
    d, cfg, sd = bending_fixture()
    model = build_model(cfg, sd).train()
    norms = {}
    load_bending_norms(model, d, norms)
    g = O.synthetic_graph(32, 16, seed=int(d4["graph_seed"]), lx=0.08, ly=0.01, with_target=True)
    # through the public API: loss() + backward()
    ls, grads_api = run_loss_backward(model, graph_to(g, DEV))
    for a, b in zip(ls, d4["loss"]):
        assert abs(a - b) <= 1e-5 * abs(b), (ls, d4["loss"])
    assert len(grads_api) == 136
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.loss(sdg, cfg, norms, g)[0].backward()           # fp32 oracle; leaves `norms` as the reference run did
    ref32 = {}
    for name, l2 in zip(d4["grad_names"], d4["grad_l2"]):
        k = str(name)
        key = "grad__" + k
        ref32[k] = torch.from_numpy(d4[key]) if key in d4 else sdg[k].grad      # the fixture holds the small tensors
        assert abs(float(grads_api[k].double().norm()) - l2) <= LOOSE_TOL * l2, (k, float(grads_api[k].double().norm()), l2)
    # same step below the autograd wrapper, statistics now as the first call left them
    copy_norms_frozen(model, norms)
    loss3, grads, masks = gpu_loss_and_grads(model, g)
    assert all(torch.equal(grads[k], grads_api[k]) for k in grads)
    l64, pinned = pinned_oracle_grads(cfg, sd, norms, g, masks)
    assert abs(float(loss3.sum()) - l64[0]) <= 1e-5 * abs(l64[0])
    check_grads(grads, pinned, ref32)


@pytest.mark.parametrize("nx,ny,steps,time_dim,mat", [(24, 12, 3, 1, False), (60, 40, 2, 2, True)])
def test_loss_grads_vs_oracle(nx, ny, steps, time_dim, mat, matmul_mode):
    """Every parameter gradient against the oracle's autograd at equal normaliser statistics."""
    cfg = O.OracleConfig(node_in_dim=3 + (time_dim + 1) + 5 + (2 if mat else 0), process_steps=steps, with_mat_params=mat,
                         time_dim=time_dim)
    sd = O.init_state_dict(cfg, 5)
    model = build_model(cfg, sd).train()
    norms = O.make_norms(cfg)
    g = O.synthetic_graph(nx, ny, seed=3, time_dim=time_dim, with_target=True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ol = O.loss(sdg, cfg, norms, g)                    # accumulates all three normalisers
    ol[0].backward()
    copy_norms_frozen(model, norms)
    loss3, grads, masks = gpu_loss_and_grads(model, g)
    for c in range(3):
        assert abs(float(loss3[c]) - float(ol[c + 1].detach())) <= 1e-5 * abs(float(ol[c + 1].detach()))
    _, pinned = pinned_oracle_grads(cfg, sd, norms, g, masks)
    check_grads(grads, pinned, {k: v.grad for k, v in sdg.items()})


def test_backward_deterministic_and_optimizer_step():
    """Two backward passes give bit-identical gradients; torch.optim.Adam consumes them
    (run_train_multiscale.py:115-119, :139-142) and the loss goes down."""
    cfg = O.OracleConfig(process_steps=2)
    sd = O.init_state_dict(cfg, 1)
    model = build_model(cfg, sd).train()
    gg = graph_to(O.synthetic_graph(40, 30, seed=6, with_target=True), DEV)
    _, g1 = run_loss_backward(model, gg)
    for nrm in (model.node_features_normalizer, model.edge_features_normalizer, model.output_normalizer):
        nrm._max_accumulations = 0                     # freeze the statistics: same function both times
    l_a, g2 = run_loss_backward(model, gg)
    l_b, g3 = run_loss_backward(model, gg)
    assert l_a == l_b and all(torch.equal(g2[k], g3[k]) for k in g2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    first = None
    for _ in range(10):
        opt.zero_grad()
        ls = model.loss(gg)
        ls[0].backward()
        opt.step()
        first = float(ls[0].detach()) if first is None else first
    assert float(ls[0].detach()) < first, (first, float(ls[0].detach()))


def test_fused_adam_matches_torch_adam():
    """gnn_amd.optim.FusedAdam against torch.optim.Adam on the CPU: same gradients for 6 steps, a
    cosine schedule changing the learning rate, tensors smaller and larger than one chunk."""
    from gnn_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(0)
    shapes = [(128, 384), (128,), (3, 128), (1,), (40000,)]
    ref = [torch.randn(*sh, generator=gen).requires_grad_(True) for sh in shapes]
    mine = [p.detach().clone().to(DEV).requires_grad_(True) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-5)
    o_mine = FusedAdam(mine, lr=1e-2, weight_decay=1e-5)
    s_ref = torch.optim.lr_scheduler.CosineAnnealingLR(o_ref, T_max=6, eta_min=1e-6)
    s_mine = torch.optim.lr_scheduler.CosineAnnealingLR(o_mine, T_max=6, eta_min=1e-6)
    for it in range(6):
        for p, q in zip(ref, mine):
            g = torch.randn(*p.shape, generator=gen) * (10.0 ** (it - 3))
            p.grad, q.grad = g.clone(), g.to(DEV)
        o_ref.step(); o_mine.step(); s_ref.step(); s_mine.step()
    for p, q in zip(ref, mine):
        assert rel_l2(q.detach(), p.detach()) < 1e-6, rel_l2(q.detach(), p.detach())
    sd = o_mine.state_dict()
    assert set(sd["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "params"}
    assert all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in sd["state"].values())


def test_fused_adam_resumes_from_a_loaded_state():
    """step, save, step, load, step: after load_state_dict the moments are new tensors; the cached pointer table
    must follow them (and not keep updating the old, freed ones).  torch.optim.Adam walks the same sequence."""
    import copy
    from gnn_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(1)
    shapes = [(128, 128), (128,), (20000,)]
    ref = [torch.randn(*sh, generator=gen).requires_grad_(True) for sh in shapes]
    mine = [p.detach().clone().to(DEV).requires_grad_(True) for p in ref]
    o_ref, o_mine = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-5), FusedAdam(mine, lr=1e-2, weight_decay=1e-5)

    def step():
        for p, q in zip(ref, mine):
            g = torch.randn(*p.shape, generator=gen)
            p.grad, q.grad = g.clone(), g.to(DEV)
        o_ref.step(); o_mine.step()

    step()
    saved_ref, saved_mine = copy.deepcopy(o_ref.state_dict()), copy.deepcopy(o_mine.state_dict())
    w_ref, w_mine = [p.detach().clone() for p in ref], [q.detach().clone() for q in mine]
    step(); step()
    with torch.no_grad():                        # roll back: weights and optimizer state of the checkpoint
        for p, w in zip(ref, w_ref):
            p.copy_(w)
        for q, w in zip(mine, w_mine):
            q.copy_(w)
    o_ref.load_state_dict(saved_ref); o_mine.load_state_dict(saved_mine)
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(4)]      # reuse whatever memory was freed
    step(); step()
    del junk
    for p, q in zip(ref, mine):
        assert rel_l2(q.detach(), p.detach()) < 1e-6, rel_l2(q.detach(), p.detach())
    st = o_mine.state[mine[0]]
    assert st["step"] == 3 and bool(torch.isfinite(st["exp_avg"]).all())


def test_fused_edge_backward_matches_unfused():
    """The training step that recomputes the edge MLP's hidden activations in the backward
    (mgn_train_edge_bwd) against the step that keeps them: same loss, gradients equal to 2e-6
    (the weight-gradient sums are grouped differently), on a mesh with ragged in-degrees."""
    import copy
    from gnn_amd import training
    cfg = O.OracleConfig(process_steps=3)
    sd = O.init_state_dict(cfg, 9)
    g = O.synthetic_graph(37, 23, seed=4)
    n = g["mesh_pos"].shape[0]
    g["target"] = torch.cat([g["world_pos"], g["phi"]], -1).unsqueeze(0) + 1e-2 * torch.randn(1, n, 3, generator=torch.Generator().manual_seed(3))
    gg = graph_to(g, DEV)
    out = []
    for fused in (True, False):
        model = build_model(cfg, sd)
        tgt = model._normalized_target(gg)
        loss3, saved = training.forward_saving(model, gg, tgt, fused=fused)
        assert saved["fused"] == fused
        out.append((loss3.cpu(), {k: v.cpu() for k, v in training.backward(model, saved, torch.ones(3, device=DEV)).items()}))
    (la, ga), (lb, gb) = out
    assert rel_l2(la, lb) < 1e-6
    worst = max((rel_l2(ga[k], gb[k]), k) for k in gb)
    assert worst[0] < 2e-6, worst


@pytest.mark.parametrize("options", [False, True], ids=["default", "attention+multiscale"])
def test_one_reduction_launch_per_step_changes_nothing(options, monkeypatch):
    """All weight-gradient partial sums of a backward pass reduced by one launch at its end (the default) against one
    small launch per call (MGN_TRAIN_DEFER=0): every gradient bit for bit, also on the second step (the partial-sum
    pool is reused), and through the C entry points directly with a record list that is too short."""
    import ctypes as C
    from gnn_amd import _lib, ops, training
    if options:
        from test_gpu_options import build
        cfg = O.OracleConfig(process_steps=2, attention=True, sample_ratio=0.4, coarse_process_steps=2)
    else:
        build = build_model
        cfg = O.OracleConfig(process_steps=3)
    sd = O.init_state_dict(cfg, 9)
    g = O.synthetic_graph(37, 23, seed=4)
    n = g["mesh_pos"].shape[0]
    g["target"] = torch.cat([g["world_pos"], g["phi"]], -1).unsqueeze(0) + 1e-2 * torch.randn(1, n, 3, generator=torch.Generator().manual_seed(3))
    gg = graph_to(g, DEV)
    out = {}
    for defer in ("1", "0"):
        monkeypatch.setenv("MGN_TRAIN_DEFER", defer)
        model = build(cfg, sd)
        model.coarse_start = 3
        steps = []
        for _ in range(2):
            tgt = model._normalized_target(gg)
            loss3, saved = training.forward_saving(model, gg, tgt)
            steps.append({k: v.clone() for k, v in training.backward(model, saved, torch.ones(3, device=DEV)).items()})
        out[defer] = steps
    for a, b in zip(out["1"], out["0"]):
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    # a list with room for one record: the second call reduces at once and reports the full list
    d = ops.DeferredReductions(DEV)
    d.recs = (_lib.MgnReduceRec * 1)()
    gen = torch.Generator().manual_seed(1)
    G, A = torch.randn(500, 128, generator=gen).to(DEV), torch.randn(500, 128, generator=gen).to(DEV)
    dw1, dw2, dw_ref = (torch.empty(128, 128, device=DEV) for _ in range(3))
    ops.train_linear_bwd(G, torch.empty(ops.train_scratch_bytes(), dtype=torch.uint8, device=DEV), a=A, dw=dw_ref)
    _lib.check(_lib.lib().mgn_train_defer_reductions(d.recs, 1, C.byref(d.count)), "defer")
    try:
        ops.train_linear_bwd(G, d, a=A, dw=dw1)
        with pytest.raises(RuntimeError, match="list is full"):
            ops.train_linear_bwd(G, d, a=A, dw=dw2)
    finally:
        _lib.lib().mgn_train_defer_reductions(None, 0, None)
    assert d.count.value == 1 and torch.equal(dw2, dw_ref)             # reduced at once all the same
    d.flush()
    assert torch.equal(dw1, dw_ref)


def test_full_size_training_step_properties():
    """The training configuration at its full size (500 x 500 nodes, 10 message-passing steps), where
    the oracle's autograd is out of reach, through properties that do not depend on size:
    the loss of the training path equals the loss formed from the inference path's ``forward``
    (different kernels for every stage of the processor), and the gradient's directional
    derivative matches a central finite difference of the loss along a random direction."""
    import gnn_amd
    from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale
    g = gnn_amd.synthetic_frames(499, 499, n_frames=1, seed=0)[0]
    n = g["mesh_pos"].shape[0]
    gen = torch.Generator().manual_seed(1)
    g["target"] = torch.cat([g["world_pos"], g["phi"]], -1).unsqueeze(0) + 1e-3 * torch.randn(1, n, 3, generator=gen)
    g = g.to(DEV)
    torch.manual_seed(0)
    model = EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=10, coarse_process_steps=0, node_out_dim=3,
                                          attention=False, with_mat_params=False, sample_ratio=0.0, time_dim=1,
                                          device=DEV).to(DEV)
    total, ux, uy, ph = model.loss(g)
    total.backward()
    grads = [p.grad.clone() for p in model.parameters()]
    assert all(torch.isfinite(x).all() for x in grads)
    # (1) the same loss from the inference kernels (statistics now hold the same data twice: same mean / std)
    with torch.no_grad():
        delta = model.forward(g)                                             # [1, N, 3] normalised prediction
        tgt = model.output_normalizer(g["target"] - torch.cat([g["world_pos"], g["phi"]], -1), accumulate=False)
        err = (delta - tgt) ** 2
        nt = g["node_type"]
        comps = [(err[:, nt[:, k + 1] != 1, k].sum(dim=1)).mean() for k in range(3)]
        ref_total = comps[0] + comps[1] + comps[2]
    assert abs(float(total.detach()) - float(ref_total)) <= 2e-5 * abs(float(ref_total)), (float(total.detach()), float(ref_total))
    # (2) directional derivative; the direction scales with each tensor so that every layer takes part
    gen = torch.Generator(device=DEV).manual_seed(5)
    direction = [torch.randn(p.shape, device=DEV, generator=gen) * (p.detach().abs().mean() + 1e-3) for p in model.parameters()]
    analytic = sum(float((gr.double() * d.double()).sum()) for gr, d in zip(grads, direction))
    eps = 2e-3
    vals = []
    with torch.no_grad():
        for sign in (1.0, -1.0):
            for p, d in zip(model.parameters(), direction):
                p.add_(d, alpha=sign * eps)
            vals.append(float(model.loss(g)[0]))
            for p, d in zip(model.parameters(), direction):
                p.sub_(d, alpha=sign * eps)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(numeric - analytic) <= 3e-2 * abs(analytic), (numeric, analytic, vals)


def test_loss_grads_irregular_graph():
    """Gradients on a directed, asymmetric graph with isolated nodes and a hub whose 300 in-edges span
    several 32-edge tiles (the receiver-segmented sums of the fused edge backward chain across tiles
    there; the sender-side sums see senders without any edge): against the fp64 oracle with the
    step's ReLU pattern pinned, and fused against unfused."""
    from gnn_amd import training
    cfg = O.OracleConfig(process_steps=2)
    sd = O.init_state_dict(cfg, 12)
    g = O.synthetic_graph(20, 15, seed=5, with_target=True)
    n = g["mesh_pos"].shape[0]
    gen = torch.Generator().manual_seed(11)
    e = 2500
    snd = torch.randint(0, n, (e,), generator=gen)
    rcv = torch.randint(0, n, (e,), generator=gen)
    rcv[:300] = 7                                                    # hub
    iso = torch.tensor([3, 50, 51, 52, n - 1])                       # nodes that neither send nor receive
    keep = ~(torch.isin(snd, iso) | torch.isin(rcv, iso)) & (snd != rcv)
    g["edge_index"] = torch.unique(torch.stack([snd[keep], rcv[keep]]), dim=1)
    model = build_model(cfg, sd).train()
    norms = O.make_norms(cfg)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ol = O.loss(sdg, cfg, norms, g)
    ol[0].backward()
    copy_norms_frozen(model, norms)
    loss3, grads, masks = gpu_loss_and_grads(model, g)
    for c in range(3):
        assert abs(float(loss3[c]) - float(ol[c + 1].detach())) <= 1e-5 * abs(float(ol[c + 1].detach()))
    _, pinned = pinned_oracle_grads(cfg, sd, norms, g, masks)
    check_grads(grads, pinned, {k: v.grad for k, v in sdg.items()})
    gg = graph_to(g, DEV)
    tgt = model._normalized_target(gg)
    _, saved = training.forward_saving(model, gg, tgt, fused=False)
    unfused = training.backward(model, saved, torch.ones(3, device=DEV))
    worst = max((rel_l2(grads[k], unfused[k].cpu()), k) for k in grads)
    assert worst[0] < 2e-6, worst


def test_fused_adam_capturable_equals_plain():
    """FusedAdam(capturable=True) (scalars through a device record, table through a pinned buffer) gives the bits of
    the plain optimiser over 5 steps with a changing learning rate."""
    from gnn_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(0)
    shapes = [(128, 384), (128,), (40000,)]
    a = [torch.randn(*sh, generator=gen).to(DEV).requires_grad_(True) for sh in shapes]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    oa, ob = FusedAdam(a, lr=1e-2, weight_decay=1e-5), FusedAdam(b, lr=1e-2, weight_decay=1e-5, capturable=True)
    for it in range(5):
        for p, q in zip(a, b):
            g = torch.randn(*p.shape, generator=gen).to(DEV)
            p.grad, q.grad = g.clone(), g.clone()
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 1e-2 / (it + 1)
            o.step()
    for p, q in zip(a, b):
        assert torch.equal(p.detach(), q.detach())
    assert oa.state[a[0]]["step"] == ob.state[b[0]]["step"] == 5


@pytest.mark.parametrize("options", [False, True], ids=["default", "attention+multiscale"])
def test_captured_train_step_equals_eager_sequence(options):
    """run_train_multiscale.py:137-142 on a small mesh, seven frames with different noise and a learning-rate change:
    the captured step (first call eager, second captured, then replays) leaves the same parameters, normaliser
    statistics and loss values as the eager loop, bit for bit.  Also with both options of the model on (the coarse
    graph pinned to one first sample)."""
    import gnn_amd
    from gnn_amd.optim import FusedAdam
    from gnn_amd.training import CapturedTrainStep
    cfg = O.OracleConfig(process_steps=3, attention=True, sample_ratio=0.4, coarse_process_steps=2) if options \
        else O.OracleConfig(process_steps=3)
    sd = O.init_state_dict(cfg, 4)
    frames = gnn_amd.synthetic_frames(14, 9, n_frames=8, seed=3)
    gen = torch.Generator().manual_seed(5)
    batch = []
    for t in range(7):
        fr = gnn_amd.MeshGraph(**dict(frames[t].items()))
        fr["swelling_phi"] = torch.stack([frames[t]["swelling_phi"], frames[t + 1]["swelling_phi"]]).T.contiguous()
        fr["target"] = torch.cat([frames[t + 1]["world_pos"], frames[t + 1]["phi"]], -1).unsqueeze(0)
        fr["world_pos"] = fr["world_pos"] + 1e-3 * torch.randn(fr["world_pos"].shape, generator=gen)
        batch.append(fr)

    def run(captured):
        if options:
            from test_gpu_options import build
            model = build(cfg, sd).train()
            model.coarse_start = 5
        else:
            model = build_model(cfg, sd).train()
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=captured)
        stepper = CapturedTrainStep(model, opt) if captured else None
        losses = []
        for it, fr in enumerate(batch):
            if it == 4:
                opt.param_groups[0]["lr"] = 3e-4
            if captured:
                losses.append(torch.stack(stepper(fr)))
            else:
                opt.zero_grad()
                out = model.loss(fr.to(DEV))
                out[0].backward()
                opt.step()
                losses.append(torch.stack([o.detach() for o in out]))
        if captured:
            assert stepper.graph is not None            # replays happened
            stepper.close()
        return model, opt, torch.stack(losses)

    m_e, o_e, l_e = run(False)
    m_c, o_c, l_c = run(True)
    assert torch.equal(l_e, l_c), (l_e, l_c)
    for (k, p), (_, q) in zip(m_e.named_parameters(), m_c.named_parameters()):
        assert torch.equal(p, q), k
    pairs = [(m_e.node_features_normalizer, m_c.node_features_normalizer), (m_e.edge_features_normalizer, m_c.edge_features_normalizer),
             (m_e.output_normalizer, m_c.output_normalizer)]
    if options:
        pairs.append((m_e.coarse_edge_features_normalizer, m_c.coarse_edge_features_normalizer))
    for a, b in pairs:
        assert a._acc_count == b._acc_count and a._num_accumulations == b._num_accumulations == 7
        assert torch.equal(a._acc_sum, b._acc_sum) and torch.equal(a._acc_sum_squared, b._acc_sum_squared)
    assert o_e.state[next(m_e.parameters())]["step"] == o_c.state[next(m_c.parameters())]["step"] == 7


def test_captured_train_step_notices_another_mesh_of_equal_shape():
    """Frames of a second mesh with the same node and edge counts but other connectivity (the first mesh with its
    nodes renumbered) must not replay the step captured on the first: after the switch the captured stepper and the
    eager loop still agree bit for bit, and the stepper has re-captured on the new mesh.  The first mesh's tensors are
    dropped before the second is built, so that the allocator is free to reuse their addresses."""
    import gc
    import numpy as np
    import gnn_amd
    from gnn_amd import reorder as R
    from gnn_amd.optim import FusedAdam
    from gnn_amd.training import CapturedTrainStep
    cfg = O.OracleConfig(process_steps=2)
    sd = O.init_state_dict(cfg, 4)

    def make_batch(shuffle):
        frames = gnn_amd.synthetic_frames(14, 9, n_frames=5, seed=3)
        if shuffle is not None:
            frames = [R.permute_frame(fr, shuffle) for fr in frames]
        out = []
        for t in range(4):
            fr = gnn_amd.MeshGraph(**dict(frames[t].items()))
            fr["swelling_phi"] = torch.stack([frames[t]["swelling_phi"], frames[t + 1]["swelling_phi"]]).T.contiguous()
            fr["target"] = torch.cat([frames[t + 1]["world_pos"], frames[t + 1]["phi"]], -1).unsqueeze(0)
            out.append(fr.to(DEV))                     # device tensors: their addresses are what a stale key would compare
        return out

    n_nodes = 15 * 10
    shuffle = np.random.default_rng(1).permutation(n_nodes)

    def run(captured):
        model = build_model(cfg, sd).train()
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=captured)
        stepper = CapturedTrainStep(model, opt) if captured else None
        losses, recaptured = [], False
        for which in (None, shuffle):
            batch = make_batch(which)
            first_graph = stepper.graph if captured else None
            for fr in batch:
                if captured:
                    losses.append(torch.stack(stepper(fr)))
                else:
                    opt.zero_grad()
                    out = model.loss(fr)
                    out[0].backward()
                    opt.step()
                    losses.append(torch.stack([o.detach() for o in out]))
            if captured and which is not None:
                recaptured = stepper.graph is not None and stepper.graph is not first_graph
            del batch, fr
            gc.collect()
            torch.cuda.empty_cache()
        if captured:
            assert recaptured
            stepper.close()
        return model, torch.stack(losses)

    m_e, l_e = run(False)
    m_c, l_c = run(True)
    assert torch.equal(l_e, l_c), (l_e, l_c)
    for (k, p), (_, q) in zip(m_e.named_parameters(), m_c.named_parameters()):
        assert torch.equal(p, q), k


def test_training_step_is_additive_over_disconnected_meshes():
    """A size-independent property at the size of the training configuration (500 x 500 nodes, here twice, three
    blocks): the loss sums over nodes, so on the block-diagonal union of a frame A and a different frame B the loss and
    every gradient equal the sums of the two separate runs (frozen statistics).  Loss to 1e-5; gradients to 1e-4 of the
    larger part (measured 1.7e-5: the weight gradients add 0.5M and 3M rows in another grouping, and the second copy's edge
    tiles start at another offset, which reorders the summands of its receiver sums)."""
    import gnn_amd
    cfg = O.OracleConfig(process_steps=3)
    sd = O.init_state_dict(cfg, 2)
    model = build_model(cfg, sd).train()
    norms = O.make_norms(cfg)
    small = O.synthetic_graph(20, 12, seed=1, with_target=True)
    O.loss({k: v.clone() for k, v in sd.items()}, cfg, norms, small)
    copy_norms_frozen(model, norms)
    frames = gnn_amd.synthetic_frames(499, 499, n_frames=3, seed=0)
    gen = torch.Generator().manual_seed(3)

    def with_target(fr, nxt):
        out = gnn_amd.MeshGraph(**dict(fr.items()))
        out["swelling_phi"] = torch.stack([fr["swelling_phi"], nxt["swelling_phi"]]).T.contiguous()
        out["target"] = torch.cat([nxt["world_pos"], nxt["phi"]], -1).unsqueeze(0)
        out["world_pos"] = out["world_pos"] + 1e-3 * torch.randn(out["world_pos"].shape, generator=gen)
        return out

    a, b = with_target(frames[0], frames[1]), with_target(frames[1], frames[2])
    la, ga = run_loss_backward(model, a.to(DEV))
    lb, gb = run_loss_backward(model, b.to(DEV))
    lu, gu = run_loss_backward(model, gnn_amd.batch_graphs([a, b]).to(DEV))
    for c in range(4):
        assert abs(lu[c] - (la[c] + lb[c])) <= 1e-5 * abs(lu[c]), (c, lu[c], la[c], lb[c])
    worst = 0.0
    for k in gu:
        scale = max(float(ga[k].double().norm()), float(gb[k].double().norm()), 1e-30)
        worst = max(worst, float((gu[k].double() - ga[k].double() - gb[k].double()).norm()) / scale)
    print("union vs sum of the parts, worst gradient deviation relative to the larger part: %.2e" % worst)
    assert worst < 1e-4
