"""GPU: the mesh-partitioned HIP path on ONE device.  Each "rank" is a Python thread with its
own model and HaloPlan; the plan's transport is replaced by an in-process exchange, everything
else (sliced kernel launches over owned rows, halo rows of the sender projections, normaliser
partial sums, state feedback) is the code that runs under torch.distributed.  The stitched
result must equal the single-GPU result bit for bit: the segment sums add in CSR order and
LayerNorm / products are per row, so the partition does not change any arithmetic."""
import threading

import pytest
import torch

import oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Bus:
    def __init__(self, world):
        self.world, self.barrier, self.slots = world, threading.Barrier(world), [None] * world


def _thread_plan_class(bus):
    """HaloPlan whose transport is an in-process exchange between the threads of this test.  Only the two transport
    primitives are replaced; packing, unpacking, the add-back and the staging of the state run in the library's kernels
    through the plan's own buffers, as under torch.distributed."""
    from gnn_amd.partition import HaloPlan

    class ThreadPlan(HaloPlan):
        def _all_to_all(self, recv, send, recv_counts, send_counts, async_op):
            """With async_op the received rows are only filled in by wait(), i.e. after the launches the caller puts in
            between: anything that reads them too early sees NaN and the comparison below fails."""
            plan = self

            def run():
                bus.slots[plan.rank] = (send, send_counts)
                bus.barrier.wait()
                off = 0
                for q in range(plan.world_size):
                    if q == plan.rank:
                        continue
                    s, counts = bus.slots[q]
                    start, cnt = sum(counts[: plan.rank]), counts[plan.rank]
                    assert cnt == recv_counts[q]
                    recv[off: off + cnt] = s[start: start + cnt]
                    off += cnt
                assert off == recv.shape[0]
                bus.barrier.wait()                    # nobody packs into its send buffer again before all have copied

            if not async_op:
                run()
                return None

            class Pending:
                def wait(self_inner):
                    run()
            recv.fill_(float("nan"))                  # stale rows must not be read before wait()
            return Pending()

        def _all_gather(self, parts, local_rows):
            bus.slots[self.rank] = local_rows
            bus.barrier.wait()
            for q in range(self.world_size):
                parts[q].copy_(bus.slots[q])
            bus.barrier.wait()

        def all_reduce_(self, t):
            bus.slots[self.rank] = t.clone()
            bus.barrier.wait()
            tot = torch.stack([bus.slots[q] for q in range(self.world_size)]).sum(0)   # same order on every rank
            bus.barrier.wait()
            t.copy_(tot)
            return t

    return ThreadPlan


def _thread_plan(ei, n, world, rank, bus):
    return _thread_plan_class(bus)(ei, n, world, rank).to(DEV)


@pytest.mark.parametrize("world,nx,ny,attention,ratio", [(2, 60, 45, False, 0.0), (3, 60, 45, False, 0.0), (4, 300, 260, False, 0.0),
                                                         (3, 60, 45, True, 0.0), (2, 60, 45, False, 0.5), (3, 50, 37, True, 0.3),
                                                         (4, 30, 21, False, 0.02)])
def test_partitioned_rollout_matches_single_gpu(world, nx, ny, attention, ratio):
    """Also with the attention option (the projected rows of the halo nodes travel like the sender projections) and with
    the coarse multiscale branch: every rank builds the same coarse graph from the gathered mesh coordinates and runs
    its share of it on a second plan, whose halo is whatever the nearest-neighbour edges reach (the last case leaves
    ranks with one or no coarse node)."""
    from gnn_amd import partition
    from gnn_amd.graph import MeshGraph
    if attention or ratio > 0:
        from test_gpu_options import build

        def build_model(cfg, sd):
            m = build(cfg, sd)
            m.coarse_start = 11
            return m
    else:
        from test_gpu_model import build_model
    cfg = O.OracleConfig(process_steps=4, attention=attention, sample_ratio=ratio, coarse_process_steps=2 if ratio > 0 else 0)
    sd = O.init_state_dict(cfg, 3)
    g = O.synthetic_graph(nx, ny, seed=5)          # the largest case: 78k nodes, every launch spans many workgroups
    n = g["mesh_pos"].shape[0]
    steps = 3

    def run_steps(model, state):
        outs = []
        for _ in range(steps):
            pred = model.predict(state)
            last = pred[-1]
            state["world_pos"], state["phi"] = last[:, :2].contiguous(), last[:, 2:].contiguous()
            outs.append(pred)
        return torch.cat(outs)

    ref = run_steps(build_model(cfg, sd), MeshGraph(**g).to(DEV))

    bus = _Bus(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            plan = _thread_plan(g["edge_index"].numpy(), n, world, rank, bus)
            model = partition.attach(build_model(cfg, sd), plan)
            gl = plan.local_graph(MeshGraph(**g)).to(DEV)
            results[rank] = run_steps(model, gl)
            if ratio > 0:                 # an unpinned first sample is rank 0's draw on every rank: runs, finite
                model.coarse_start = None
                assert bool(torch.isfinite(model.predict(gl)).all())
        except Exception as exc:          # noqa: BLE001
            errors.append(exc)
            bus.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=100) for t in threads]
    assert not errors, errors
    out = torch.cat(results, dim=1)
    assert out.shape == ref.shape
    # the float64 normaliser partial sums are added in a different grouping than on one GPU,
    # which can move the fp32 statistics by one ulp; everything else is bit-identical
    assert rel_l2(out, ref) < 1e-6, rel_l2(out, ref)


@pytest.mark.parametrize("world,attention,ratio", [(2, False, 0.0), (3, False, 0.0), (3, True, 0.0), (2, False, 0.5), (3, True, 0.3)])
def test_partitioned_training_step_matches_single_gpu(world, attention, ratio):
    """loss + backward on a mesh split over ranks: forward halo of the sender projections, reverse
    halo of their gradients, one all-reduce of the loss and of the parameter gradients.  Every rank
    must end with the single-GPU loss and gradients (sums are grouped differently: 1e-5).  With the coarse
    branch the same happens once more on the coarse graph's own plan."""
    from gnn_amd import partition
    from gnn_amd.graph import MeshGraph
    if attention or ratio > 0:       # the attn_lin rows of the halo senders travel forward, their gradients back
        from test_gpu_options import build

        def build_model(cfg, sd):
            m = build(cfg, sd).train()
            m.coarse_start = 11
            return m
    else:
        from test_gpu_model import build_model
    cfg = O.OracleConfig(process_steps=3, attention=attention, sample_ratio=ratio, coarse_process_steps=2 if ratio > 0 else 0)
    sd = O.init_state_dict(cfg, 11)
    g = O.synthetic_graph(40, 33, seed=8)
    n = g["mesh_pos"].shape[0]
    gen = torch.Generator().manual_seed(2)
    g["target"] = (torch.cat([g["world_pos"], g["phi"]], dim=-1)
                   + 0.01 * torch.randn(cfg.time_dim, n, 3, generator=gen))

    def loss_and_grads(model, graph):
        # what loss() + backward() run, called directly: torch's autograd engine would run the
        # backward of all "ranks" on its one worker thread for the device, where the in-process
        # exchanges of this test would wait for each other for ever
        from gnn_amd import training
        tgt = model._normalized_target(graph)
        loss3, saved = training.forward_saving(model, graph, tgt)
        grads = training.backward(model, saved, torch.ones(3, device=DEV))
        return loss3.clone(), {k: v.clone() for k, v in grads.items()}

    ref_model = build_model(cfg, sd)
    ref_loss, ref_grads = loss_and_grads(ref_model, MeshGraph(**g).to(DEV))

    bus = _Bus(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            plan = _thread_plan(g["edge_index"].numpy(), n, world, rank, bus)
            model = partition.attach(build_model(cfg, sd), plan)
            gl = plan.local_graph(MeshGraph(**g)).to(DEV)
            results[rank] = loss_and_grads(model, gl) + (model.output_normalizer._acc_count,)
        except Exception as exc:          # noqa: BLE001
            errors.append(exc)
            bus.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=100) for t in threads]
    assert not errors, errors
    for loss, grads, count in results:
        assert count == ref_model.output_normalizer._acc_count
        assert rel_l2(loss, ref_loss) < 1e-5, rel_l2(loss, ref_loss)
        worst = max(rel_l2(grads[k], ref_grads[k]) for k in ref_grads)
        assert worst < 1e-5, worst


def test_million_node_mesh_cut_into_eight_ranks():
    """The largest partitioned configuration at its stated size: the 1000 x 1000-node mesh (5 992 002 edges, 15
    blocks) cut into 8 node ranges, every rank building only its own strip.  (i) Two rollout steps with fresh,
    accumulating statistics: the stitched predictions equal the single-GPU run (1e-6: per-step state exchange,
    all-reduced normaliser sums, 15 halo exchanges per step).  (ii) With frozen statistics the prediction on
    a 64 x 64-node window that straddles the cut between ranks 2 and 3 equals the CPU oracle run on that window alone,
    at every node further than 15 hops from the window's cut sides (1e-5)."""
    import gnn_amd
    from gnn_amd import partition
    from gnn_amd.graph import MeshGraph
    from test_gpu_model import build_model, copy_norms_frozen
    side, world, steps, crop = 1000, 8, 15, 64
    cfg = O.OracleConfig(process_steps=steps)
    sd = O.init_state_dict(cfg, 7)
    full, all_edges, _, n, e = gnn_amd.synthetic_strip(side - 1, side - 1, 1, 0, seed=0)
    assert n == side * side and e == all_edges.shape[1] == 5_992_002
    full["edge_index"] = torch.from_numpy(all_edges)

    # window: columns 0..63 (the left side is the mesh's own boundary), rows around the first row of rank 3
    row3 = partition.node_bounds(n, world)[3] // side
    y0 = row3 - crop // 2
    ids = torch.arange(n)
    inside = ((ids % side) < crop) & ((ids // side) >= y0) & ((ids // side) < y0 + crop)
    new_id = torch.full((n,), -1, dtype=torch.long)
    new_id[inside] = torch.arange(int(inside.sum()))
    ei = full["edge_index"]
    keep = inside[ei[0]] & inside[ei[1]]
    gc = {k: full[k][inside] for k in ("mesh_pos", "world_pos", "phi", "swelling_phi", "node_type")}
    gc["mat_param"], gc["edge_index"] = full["mat_param"], new_id[ei[:, keep]]
    norms = O.make_norms(cfg)
    O.forward(sd, cfg, norms, gc)                          # statistics from the window, then frozen on both sides
    norms["output"].normalize(1e-3 * torch.randn(1, crop * crop, 3, generator=torch.Generator().manual_seed(4)))
    for v in norms.values():
        v.max_accumulations = 0
    ref_crop = O.predict(sd, cfg, norms, gc)

    def run(model, state, n_steps):
        outs = []
        for _ in range(n_steps):
            pred = model.predict(state)
            last = pred[-1]
            state["world_pos"], state["phi"] = last[:, :2].contiguous(), last[:, 2:].contiguous()
            outs.append(pred)
        return torch.cat(outs)

    def frozen_model():
        m = build_model(cfg, sd)
        copy_norms_frozen(m, norms)
        return m

    fields = {k: v for k, v in full.items() if k != "node_ids"}
    ref_fresh = run(build_model(cfg, sd), MeshGraph(**fields).to(DEV), 2).cpu()
    ref_frozen = run(frozen_model(), MeshGraph(**fields).to(DEV), 1).cpu()

    bus = _Bus(world)
    Plan = _thread_plan_class(bus)
    fresh, frozen, report, errors = [None] * world, [None] * world, [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            frame, in_e, out_e, n_tot, e_tot = gnn_amd.synthetic_strip(side - 1, side - 1, world, rank, seed=0)
            plan = Plan.from_local(in_e, out_e, n_tot, e_tot, world, rank).to(DEV)
            report[rank] = (plan.num_owned, plan.num_halo, plan.num_send, plan.topology.num_edges, plan.interior[2],
                            sum(r[2] for r in plan.boundary))
            gl = gnn_amd.strip_local_graph(frame, plan)
            fresh[rank] = run(partition.attach(build_model(cfg, sd), plan), gl.to(DEV), 2).cpu()
            frozen[rank] = run(partition.attach(frozen_model(), plan), gl.to(DEV), 1).cpu()
        except Exception as exc:          # noqa: BLE001
            errors.append(exc)
            bus.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    assert not errors, errors
    print("rank: owned halo sent edges interior-edges boundary-edges")
    for r, rep in enumerate(report):
        print(r, *rep)
    assert sum(rep[0] for rep in report) == n and sum(rep[3] for rep in report) == e
    out_fresh, out_frozen = torch.cat(fresh, dim=1), torch.cat(frozen, dim=1)
    assert out_fresh.shape == ref_fresh.shape and bool(torch.isfinite(out_fresh).all())
    assert rel_l2(out_fresh, ref_fresh) < 1e-6, rel_l2(out_fresh, ref_fresh)
    cur = torch.cat([full["world_pos"], full["phi"]], -1)
    assert rel_l2(out_fresh[:1] - cur, ref_fresh[:1] - cur) < 5e-5           # the steps themselves, not only the states
    assert rel_l2(out_frozen, ref_frozen) < 1e-6, rel_l2(out_frozen, ref_frozen)
    # (ii) the window across the cut against the CPU oracle
    cx, cy = torch.arange(crop * crop) % crop, torch.arange(crop * crop) // crop
    far = (cx < crop - steps - 1) & (cy > steps) & (cy < crop - steps - 1)
    owner = torch.bucketize(ids[inside][far], torch.tensor(partition.node_bounds(n, world)[1:]), right=True)
    assert set(owner.tolist()) == {2, 3}                                    # both sides of the cut are checked
    got = out_frozen[:, inside][:, far]
    cur_c = torch.cat([gc["world_pos"], gc["phi"]], -1)[far]
    assert rel_l2(got, ref_crop[:, far]) < 1e-5
    assert rel_l2(got - cur_c, ref_crop[:, far] - cur_c) < 5e-5, rel_l2(got - cur_c, ref_crop[:, far] - cur_c)
