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


def _thread_plan(ei, n, world, rank, bus):
    from gnn_amd.partition import HaloPlan

    class ThreadPlan(HaloPlan):
        def exchange_rows(self, buf, async_op=False):
            """In-process transport.  With async_op the halo rows are only filled in by wait(), i.e.
            after the launches the caller puts in between: anything that reads them too early sees
            the previous contents and the comparison below fails."""
            plan = self

            def run():
                send = buf.index_select(0, plan.send_idx)
                bus.slots[plan.rank] = (send, plan.send_counts)
                bus.barrier.wait()
                off = plan.num_owned
                for q in range(plan.world_size):
                    if q == plan.rank:
                        continue
                    s, counts = bus.slots[q]
                    start, cnt = sum(counts[: plan.rank]), counts[plan.rank]
                    buf[off: off + cnt] = s[start: start + cnt]
                    off += cnt
                bus.barrier.wait()

            if not async_op:
                run()
                return buf

            class Pending:
                def wait(self_inner):
                    run()
            buf[plan.num_owned:].fill_(float("nan"))      # stale halo rows must not be read before wait()
            return Pending()

        def return_rows(self, buf, async_op=False):
            plan = self

            def run():
                bus.slots[plan.rank] = (buf[plan.num_owned:].clone(), plan.recv_counts)
                bus.barrier.wait()
                got = []
                for q in range(plan.world_size):
                    if q == plan.rank:
                        continue
                    s, counts = bus.slots[q]                 # q's halo rows, grouped by owner
                    start, cnt = sum(counts[: plan.rank]), counts[plan.rank]
                    got.append(s[start: start + cnt])
                bus.barrier.wait()
                plan._add_returned(buf, torch.cat(got) if got else buf.new_empty((0, buf.shape[1])))

            if not async_op:
                run()
                return buf

            class Pending:
                def wait(self_inner):
                    run()
            return Pending()

        def all_reduce_(self, t):
            bus.slots[self.rank] = t.clone()
            bus.barrier.wait()
            tot = torch.stack([bus.slots[q] for q in range(self.world_size)]).sum(0)
            bus.barrier.wait()
            t.copy_(tot)
            return t

        def sum_hook(self, kind):
            total = self.global_nodes if kind == "node" else self.global_edges

            def hook(sums):
                bus.slots[self.rank] = sums.clone()
                bus.barrier.wait()
                tot = torch.stack([bus.slots[q] for q in range(self.world_size)]).sum(0)   # same order on every rank
                bus.barrier.wait()
                sums.copy_(tot)
                return total
            return hook

    return ThreadPlan(ei, n, world, rank).to(DEV)


@pytest.mark.parametrize("world,nx,ny,attention", [(2, 60, 45, False), (3, 60, 45, False), (4, 300, 260, False), (3, 60, 45, True)])
def test_partitioned_rollout_matches_single_gpu(world, nx, ny, attention):
    """Also with the attention option: the projected rows of the halo nodes travel like the sender projections."""
    from gnn_amd import partition
    from gnn_amd.graph import MeshGraph
    if attention:
        from test_gpu_options import build

        def build_model(cfg, sd):
            return build(cfg, sd)
    else:
        from test_gpu_model import build_model
    cfg = O.OracleConfig(process_steps=4, attention=attention)
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


@pytest.mark.parametrize("world,attention", [(2, False), (3, False), (3, True)])
def test_partitioned_training_step_matches_single_gpu(world, attention):
    """loss + backward on a mesh split over ranks: forward halo of the sender projections, reverse
    halo of their gradients, one all-reduce of the loss and of the parameter gradients.  Every rank
    must end with the single-GPU loss and gradients (sums are grouped differently: 1e-5)."""
    from gnn_amd import partition
    from gnn_amd.graph import MeshGraph
    if attention:       # the attn_lin rows of the halo senders travel forward, their gradients back
        from test_gpu_options import build

        def build_model(cfg, sd):
            return build(cfg, sd).train()
    else:
        from test_gpu_model import build_model
    cfg = O.OracleConfig(process_steps=3, attention=attention)
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
