#!/usr/bin/env python3
"""Training-step benchmark (loss + backward, optimizer step timed separately) on a synthetic
square mesh: the 250k-node / 10 message-passing-step training configuration by default.

    python scripts/bench_train.py [--grid 500] [--mp-steps 10] [--steps 5] [--warmup 2] [--cpu-grid 100]

Under ``python -m torch.distributed.run --nproc-per-node N`` the mesh is grid x (N * grid) nodes cut into
N strips (one rank each, RCCL; MGN_DIST_BACKEND=gloo rehearses it with ranks sharing GPUs).

Prints one JSON line.  The CPU figure is the oracle's autograd (torch CPU, all host threads) on a
smaller mesh, scaled by edge count."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=500)
    ap.add_argument("--mp-steps", type=int, default=10)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-grid", type=int, default=100, help="side of the CPU baseline mesh (0 = skip)")
    args = ap.parse_args()
    import gnn_amd
    from gnn_amd import ops
    from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MGN_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    g = gnn_amd.synthetic_frames(args.grid - 1, args.grid * world - 1, n_frames=1, seed=0)[0]
    n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
    gen = torch.Generator().manual_seed(1)
    g["target"] = torch.cat([g["world_pos"], g["phi"]], -1).unsqueeze(0) + 1e-3 * torch.randn(1, n, 3, generator=gen)
    plan = None
    if world > 1:
        from gnn_amd import partition
        plan = partition.HaloPlan(g["edge_index"].numpy(), n, world, rank).to(dev)
        g = plan.local_graph(g)
    g = g.to(dev)
    torch.manual_seed(0)
    model = EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=args.mp_steps, coarse_process_steps=0,
                                          node_out_dim=3, attention=False, with_mat_params=False, sample_ratio=0.0,
                                          time_dim=1, device=dev).to(dev).train()
    if plan is not None:
        partition.attach(model, plan)
    from gnn_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-5)

    def step(timers=None):
        t0 = time.perf_counter()
        opt.zero_grad()
        ls = model.loss(g)
        if timers is not None:
            torch.cuda.synchronize(); t1 = time.perf_counter()
        ls[0].backward()
        if timers is not None:
            torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        if timers is not None:
            torch.cuda.synchronize(); t3 = time.perf_counter()
            timers.append((t1 - t0, t2 - t1, t3 - t2))
        return ls[0]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    parts = []
    for _ in range(3):
        step(parts)
    fwd, bwd, optim = (float(np.mean([p[i] for p in parts])) for i in range(3))
    out = {"metric": "training steps/s (loss forward + backward + Adam), 1 MI355X", "value": 1.0 / dt, "unit": "steps/s",
           "ms_per_step": dt * 1e3, "ms_forward": fwd * 1e3, "ms_backward": bwd * 1e3, "ms_optimizer": optim * 1e3,
           "config": {"workload": f"synthetic {args.grid}x{args.grid}-node mesh, loss + backward of all 136 parameters",
                      "nodes": n, "edges": e, "mp_steps": args.mp_steps},
           "edge_updates_fwd_bwd_per_s": e * args.mp_steps / dt, "dtype": "f32", "loss": float(last.detach()),
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}
    if world > 1:
        out["n_gpus"], out["config"]["parallelism"] = world, f"node-range partition x{world}, {args.grid}x{args.grid} nodes per rank"
        gsum = torch.stack([p.grad.double().abs().sum() for p in model.parameters()]).sum().reshape(1).cpu()
        both = [torch.zeros_like(gsum) for _ in range(world)]
        dist.all_gather(both, gsum)
        out["grad_checksum_equal_on_all_ranks"] = bool(all(torch.equal(b, both[0]) for b in both))
        if rank != 0:
            dist.destroy_process_group()
            return
    if args.cpu_grid > 0 and world == 1:
        import oracle as O
        torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))
        cfg = O.OracleConfig(process_steps=args.mp_steps)
        sd = {k: v.clone().requires_grad_(True) for k, v in O.init_state_dict(cfg, 0).items()}
        gc = O.synthetic_graph(args.cpu_grid - 1, args.cpu_grid - 1, seed=0, with_target=True)
        norms = O.make_norms(cfg)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.loss(sd, cfg, norms, gc)[0].backward()
            ts.append(time.perf_counter() - t0)
        ec = gc["edge_index"].shape[1]
        tc = float(np.mean(ts[1:]))
        out["cpu_baseline"] = {"kind": "port", "cores": torch.get_num_threads(), "sample":
                               f"oracle loss + autograd backward on a {args.cpu_grid}x{args.cpu_grid} mesh (E={ec}), "
                               f"{tc * 1e3:.0f} ms per step", "edge_updates_fwd_bwd_per_s": ec * args.mp_steps / tc,
                               "steps_per_s_scaled_to_workload": 1.0 / (tc * e / ec)}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
