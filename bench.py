#!/usr/bin/env python3
"""Benchmark of the MeshGraphNets hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload: one autoregressive rollout step (`predict`: features ->
normalisers -> encoders -> 15 message-passing steps -> decoder -> integrator) on a synthetic
1000 x 1000-node triangle mesh (1 000 000 nodes, 5 992 002 directed edges), latent 128, fp32,
seeded random weights, inputs resident in HBM.  With N > 1 the SAME mesh is partitioned by
contiguous node ranges into N strips, one process per GPU, with a halo exchange of the sender
projections over RCCL once per message-passing step ("strong" scaling: total work fixed).
`--scaling weak` keeps 1000 x 1000 nodes per GPU instead (a 1000 x (N * 1000) mesh).  Every
rank builds only its own strip of the mesh.

`python bench.py --gpus N` from a plain shell starts the N ranks itself (a torch.distributed.run
child, before this process touches a GPU) and relays rank 0's line; under torch.distributed.run
(RANK / WORLD_SIZE in the environment) it is one of the ranks.

Prints ONE JSON line on rank 0.  `value` = edge updates per second over the whole job
= E * mp_steps * K / wall time of the K timed steps (max over ranks).  At N = 1 the line also
carries a `train` object: loss + backward + fused Adam on the 500 x 500 mesh with 10 blocks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X dense bf16 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X dense fp16 matrix peak
PEAK_HBM_GBS = 8000.0
FLOP_PER_EDGE_CANONICAL = 2 * (384 * 128 + 128 * 128)      # 131072: reference edge MLP
FLOP_PER_NODE_CANONICAL = 2 * (256 * 128 + 128 * 128)      # 98304: reference node MLP
FLOP_PER_EDGE_EDGE_KERNEL = 2 * (128 * 128 + 128 * 128)    # 65536: what mp_edge_kernel is responsible for
BYTES_PER_EDGE_EDGE_KERNEL = 512 + 512 + 8                 # e in, e out, sender + receiver ids


def pmc_record(files=("r03_pmc_traffic.json", "r02_pmc_traffic.json")):
    """The newest committed counter record among `files` (profiles/): FETCH_SIZE and WRITE_SIZE are collected in
    separate rocprofv3 --pmc runs, which a timed run cannot do itself."""
    for name in files:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            rec["file"] = "profiles/" + name
            return rec
        except OSError:
            continue
    return {}


def pmc_traffic(kernel, units, files=("r03_pmc_traffic.json", "r02_pmc_traffic.json")):
    """HBM bytes per launch of `kernel` from the committed counter pass, scaled to `units` edges (or nodes, for a
    record taken per node); None when no pass for this kernel is on file."""
    rec = pmc_record(files)
    if rec.get("kernel") != kernel:
        return None
    per = rec.get("traffic_bytes_per_edge", rec.get("traffic_bytes_per_node"))
    return None if per is None else per * units


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=1000, help="nodes per side of the square mesh")
    ap.add_argument("--mp-steps", type=int, default=15)
    ap.add_argument("--cpu-reps", type=int, default=2, help="timed repetitions of the CPU baseline layer (0 = skip)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: partition the fixed grid x grid mesh (strong) or keep grid x grid nodes per GPU (weak)")
    ap.add_argument("--reorder", choices=("none", "morton", "rcm"), default="none",
                    help="N = 1 experiment: renumber the mesh nodes before the run (the headline workload is row-major: none)")
    ap.add_argument("--strict-steps", type=int, default=2, help="N = 1: timed steps of the strict fp32 matrix path (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=3, help="N = 1: timed training steps for the `train` object (0 = skip)")
    ap.add_argument("--train-grid", type=int, default=500)
    ap.add_argument("--train-mp-steps", type=int, default=10)
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the ranks as a child job.
    Nothing here initialises a GPU (device_count() does not), so no process that has touched one is replaced."""
    backend = os.environ.get("MGN_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs (one rank per GPU over RCCL), this host shows "
                         f"{have}; MGN_DIST_BACKEND=gloo rehearses the multi-rank path with ranks sharing GPUs\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def make_model(mp_steps, device):
    from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale
    torch.manual_seed(0)
    return EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=mp_steps, coarse_process_steps=0,
                                         node_out_dim=3, attention=False, with_mat_params=False, sample_ratio=0.0,
                                         time_dim=1, device=device).to(device)


def build_problem(side, mp_steps, device, world_size, rank, scaling, reorder="none"):
    """(model, full frame or None, this rank's frame on the device, plan, global nodes, global edges, mesh rows)."""
    import gnn_amd
    from gnn_amd import partition
    model = make_model(mp_steps, device).eval()
    if world_size == 1:
        g = gnn_amd.synthetic_frames(side - 1, side - 1, n_frames=1, seed=0)[0]
        if reorder != "none":               # experiment: the same mesh with its nodes renumbered for locality
            from gnn_amd import reorder as R
            g = R.permute_frame(g, R.node_order(g, reorder))
        return model, g, g.to(device), None, g["mesh_pos"].shape[0], g["edge_index"].shape[1], side
    # every rank builds its own strip (plus one mesh row on either side), never the whole mesh
    rows = side if scaling == "strong" else side * world_size
    frame, in_e, out_e, n, e = gnn_amd.synthetic_strip(side - 1, rows - 1, world_size, rank, seed=0)
    plan = partition.HaloPlan.from_local(in_e, out_e, n, e, world_size, rank).to(device)
    partition.attach(model, plan)
    return model, None, gnn_amd.strip_local_graph(frame, plan).to(device), plan, n, e, rows


def train_object(args, device):
    """Driver-visible training figure (N = 1): loss + backward + fused Adam per step on the 250k-node mesh with
    10 blocks (the training configuration), per-kernel device time from events on the launch stream."""
    import gnn_amd
    from gnn_amd import ops
    from gnn_amd.optim import FusedAdam
    g = gnn_amd.synthetic_frames(args.train_grid - 1, args.train_grid - 1, n_frames=1, seed=0)[0]
    n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
    gen = torch.Generator().manual_seed(1)
    g["target"] = torch.cat([g["world_pos"], g["phi"]], -1).unsqueeze(0) + 1e-3 * torch.randn(1, n, 3, generator=gen)
    g = g.to(device)
    model = make_model(args.train_mp_steps, device).train()
    opt = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-5)

    def step():
        opt.zero_grad()
        ls = model.loss(g)
        ls[0].backward()
        opt.step()
        return ls[0]

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.train_steps
    log = []
    ops.set_event_log(log)
    step()
    torch.cuda.synchronize()
    ops.set_event_log(None)
    per = {}
    for name, s0, s1 in log:
        per.setdefault(name, []).append(s0.elapsed_time(s1))
    kern = {k: {"calls": len(v), "ms_total": float(np.sum(v))} for k, v in per.items()}
    out = {"metric": "training steps/s (loss forward + backward + fused Adam)", "value": 1.0 / dt, "unit": "steps/s",
           "ms_per_step": dt * 1e3, "loss": float(last.detach()),
           "config": {"workload": f"synthetic {args.train_grid}x{args.train_grid}-node mesh, {args.train_mp_steps} blocks, "
                                  "loss + backward of all 136 parameters + Adam", "nodes": n, "edges": e,
                      "mp_steps": args.train_mp_steps},
           "edge_updates_fwd_bwd_per_s": e * args.train_mp_steps / dt, "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9,
           "kernel_ms_one_step": kern}
    # dominant kernel of the backward: the fused edge backward (one launch per block).  Algorithmic bytes per
    # edge: e_in, upstream g_e read, g1 written (512 B each) + ids (8 B); gathers of ps/pr/g_aggr come from L2.
    eb = per.get("train_edge_bwd")
    if eb:
        t = float(np.mean(eb)) * 1e-3
        bytes_alg = (3 * 512 + 8) * e
        out["roofline"] = {"kernel": "train_edge_bwd_kernel", "bound": "hbm", "achieved": bytes_alg / t / 1e9,
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bytes_alg / t / 1e9 / PEAK_HBM_GBS,
                           "traffic": pmc_traffic("train_edge_bwd_kernel", e, ("r03_pmc_traffic_train.json", "r02_pmc_traffic_train.json")),
                           "traffic_unit": "bytes per launch (rocprofv3 PMC passes, profiles/r0*_pmc_traffic_train.json)",
                           "avg_launch_ms": t * 1e3, "bytes_per_edge_algorithmic": 3 * 512 + 8}
    return out


def other_configs(device, cpu=True):
    """The two single-GPU configurations of the baseline that the headline workload does not cover, each a few hundred
    milliseconds: the bending-mesh rollout (33 x 17 nodes, 10 blocks, 200 frames) and the 100k-node mesh with 15 blocks
    advanced as 8 block-diagonal trajectories (`scripts/bench_configs.py` prints the same figures)."""
    import gnn_amd
    from gnn_amd.rollout import rollout
    out = {}
    torch.manual_seed(0)
    m = make_model(10, device).eval()
    frames = gnn_amd.synthetic_frames(32, 16, n_frames=200, seed=0, lx=0.08, ly=0.01)
    rollout(m, frames[:20])
    torch.cuda.synchronize(device); t0 = time.perf_counter()
    rollout(m, frames)
    torch.cuda.synchronize(device); dt = time.perf_counter() - t0
    out["bending_rollout"] = {"workload": "33x17-node bending mesh, 10 MP steps, 200-frame rollout (upload, loop, RMSE)",
                              "rollout_steps_per_s": 200 / dt, "ms_per_step": dt / 200 * 1e3,
                              "edge_updates_per_s": 3168 * 10 * 200 / dt}
    if cpu:
        # the same step on the host cores: the CPU oracle's predict on this mesh (20 timed calls after 2 warm-up calls)
        import oracle as O
        cfg = O.OracleConfig(process_steps=10)
        sd, norms = O.init_state_dict(cfg, 0), O.make_norms(cfg)
        fr = {k: frames[0][k] for k in ("world_pos", "mesh_pos", "phi", "node_type", "mat_param", "edge_index")}
        fr["swelling_phi"] = torch.stack([frames[0]["swelling_phi"], frames[1]["swelling_phi"]]).T.contiguous()   # window t, t+1
        with torch.no_grad():
            for _ in range(2):
                O.predict(sd, cfg, norms, fr)
            t0 = time.perf_counter()
            for _ in range(20):
                O.predict(sd, cfg, norms, fr)
            dtc = (time.perf_counter() - t0) / 20
        out["bending_rollout"]["cpu_baseline"] = {"value": 1.0 / dtc, "unit": "rollout steps/s", "cores": torch.get_num_threads(),
                                                  "kind": "port", "sample": "20 predict calls of the CPU oracle on the same mesh",
                                                  "ms_per_step": dtc * 1e3}
    # training step on the same mesh (the size of the reference's own training log): eager, and replayed from a captured graph
    from gnn_amd.optim import FusedAdam
    from gnn_amd.training import CapturedTrainStep
    fr = frames[0]
    fr = gnn_amd.MeshGraph(**dict(fr.items()))
    fr["swelling_phi"] = torch.stack([frames[0]["swelling_phi"], frames[1]["swelling_phi"]]).T.contiguous()
    fr["target"] = torch.cat([frames[1]["world_pos"], frames[1]["phi"]], -1).unsqueeze(0)
    fr = fr.to(device)
    tr = {}
    for captured in (False, True):
        mt = make_model(10, device).train()
        opt = FusedAdam(mt.parameters(), lr=1e-4, weight_decay=1e-5, capturable=captured)
        stepper = CapturedTrainStep(mt, opt) if captured else None

        def one():
            if captured:
                return stepper(fr)
            opt.zero_grad()
            out = mt.loss(fr)
            out[0].backward()
            opt.step()
            return out
        for _ in range(5):
            one()
        torch.cuda.synchronize(device); t0 = time.perf_counter()
        for _ in range(50):
            one()
        torch.cuda.synchronize(device); dtt = (time.perf_counter() - t0) / 50
        tr["captured_graph" if captured else "eager"] = {"steps_per_s": 1.0 / dtt, "ms_per_step": dtt * 1e3}
        if captured:
            stepper.close()
    out["bending_training"] = {"workload": "33x17-node bending mesh, 10 MP steps: loss + backward + fused Adam per step", **tr}
    g = gnn_amd.batch_graphs([gnn_amd.synthetic_frames(316, 316, seed=0)[0]] * 8).to(device)
    m = make_model(15, device).eval()
    e = g["edge_index"].shape[1]
    for _ in range(2):
        m.predict(g)
    torch.cuda.synchronize(device); t0 = time.perf_counter()
    for _ in range(4):
        m.predict(g)
    torch.cuda.synchronize(device); dt = (time.perf_counter() - t0) / 4
    out["batched_100k_x8"] = {"workload": "317x317-node mesh, 15 MP steps, 8 trajectories as one block-diagonal graph",
                              "ms_per_step": dt * 1e3, "trajectory_steps_per_s": 8 / dt, "edge_updates_per_s": e * 15 / dt}
    if cpu:
        # one trajectory of the same mesh on the host cores: the CPU oracle's predict (1 warm-up + 1 timed call)
        import oracle as O
        cfg = O.OracleConfig(process_steps=15)
        sd, norms = O.init_state_dict(cfg, 0), O.make_norms(cfg)
        f1 = gnn_amd.synthetic_frames(316, 316, seed=0)[0]
        fr = {k: f1[k] for k in ("world_pos", "mesh_pos", "phi", "swelling_phi", "node_type", "mat_param", "edge_index")}
        with torch.no_grad():
            O.predict(sd, cfg, norms, fr)
            t0 = time.perf_counter()
            O.predict(sd, cfg, norms, fr)
            dtc = time.perf_counter() - t0
        e1 = f1["edge_index"].shape[1]
        out["batched_100k_x8"]["cpu_baseline"] = {
            "value": 1.0 / dtc, "unit": "trajectory steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 timed + 1 warm-up predict call of the CPU oracle on ONE 317x317-node trajectory (15 MP steps)",
            "ms_per_step": dtc * 1e3, "edge_updates_per_s": e1 * 15 / dtc}
    return out


def cpu_baseline(g, mp_steps, reps):
    """The CPU oracle (a port of the reference's processor block) on this host's cores: one
    message-passing layer on the full mesh, the same inputs shape as the GPU run."""
    import oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))
    cfg = O.OracleConfig(process_steps=1)
    sd = O.init_state_dict(cfg, 0)
    n, e = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
    gen = torch.Generator().manual_seed(0)
    x, ef = torch.randn(n, 128, generator=gen), torch.randn(e, 128, generator=gen)
    times = []
    with torch.no_grad():
        for i in range(reps + 1):
            t0 = time.perf_counter()
            O.mp_layer(sd, 0, x, g["edge_index"], ef)
            times.append(time.perf_counter() - t0)
    t = float(np.mean(times[1:])) if reps > 0 else times[0]
    return {"value": e / t, "unit": "edge-updates/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} timed + 1 warm-up message-passing layer(s) of the CPU oracle on the full mesh "
                      f"(N={n}, E={e}), {t * 1e3:.0f} ms per layer",
            "ms_per_mp_step": t * 1e3}


def main():
    args = parse()
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    if args.gpus != world_size:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # MGN_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks
    # share devices, halos travel through host memory); the default is RCCL, one GPU per rank
    backend = os.environ.get("MGN_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world_size > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from gnn_amd import ops
    model, g_full, g, plan, n, e, rows = build_problem(args.grid, args.mp_steps, device, world_size, rank, args.scaling, args.reorder)

    def step(state):
        pred = model.predict(state, check_range=False)      # range flag read once, after the timed region
        last = pred[-1]
        state["world_pos"], state["phi"] = last[:, :2].contiguous(), last[:, 2:].contiguous()

    def sync():
        if world_size > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(k):
        """Wall time of k steps between two barrier + synchronize points, maximum over the ranks."""
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            step(g)
        sync()
        t = time.perf_counter() - t0
        if world_size > 1:
            import torch.distributed as dist
            tt = torch.tensor([t], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    def kernel_times(log):
        per = {}
        for name, s0, s1 in log:
            per.setdefault(name, []).append(s0.elapsed_time(s1))
        return per

    for _ in range(args.warmup):
        step(g)
    # the K timed steps: the launch sequence users get, nothing recorded in between
    dt = timed_steps(args.steps)
    # the same K steps again with HIP events around every launch (on the launch stream) for the per-kernel times;
    # the launch sequence is identical, the events add host work and small gaps, so this is not the headline time
    log = []
    ops.set_event_log(log)
    dt_inst = timed_steps(args.steps)
    ops.set_event_log(None)
    mode = ops.matmul_mode()
    ops.raise_if_out_of_range("bench.py")
    finite = bool(torch.isfinite(g["world_pos"]).all())
    per = kernel_times(log)
    kern_ms = {k: float(np.mean(v)) for k, v in per.items()}

    strict = None
    if world_size == 1 and mode != "f32" and args.strict_steps > 0:
        # the same workload on the strict path: fp32 matrix instructions everywhere, no operand splitting
        ops.set_matmul_mode("f32")
        try:
            step(g)
            dts = timed_steps(args.strict_steps)
            slog = []
            ops.set_event_log(slog)
            step(g)
            torch.cuda.synchronize()
            ops.set_event_log(None)
            sper = kernel_times(slog)
            strict = {"matrix_path": "v_mfma_f32_32x32x2_f32 (MGN_MATMUL=f32)", "steps": args.strict_steps,
                      "ms_per_step": dts / args.strict_steps * 1e3,
                      "edge_updates_per_s": e * args.mp_steps * args.strict_steps / dts,
                      "rollout_steps_per_s": args.strict_steps / dts,
                      "edge_kernel_ms": float(np.mean(sper.get("mp_edge", [float("nan")]))),
                      "node_kernel_ms": float(np.mean(sper.get("mp_node", [float("nan")])))}
        finally:
            ops.set_matmul_mode(mode)

    if rank == 0:
        e_local = plan.topology.num_edges if plan is not None else e
        n_local = plan.num_owned if plan is not None else n
        e_kernel = plan.interior[2] if plan is not None else e      # edges of the launches timed as "mp_edge"
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "edge-updates/sec per MP step + rollout steps/sec",
            "value": e * args.mp_steps * args.steps / dt,
            "unit": "edge-updates/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": {"f16x2": "f32 (2-term fp16 split operands, fp32 accumulate)", "bf16x3": "f32 (3-term bf16 split operands, fp32 accumulate)",
                                                 "f32": "f32"}[mode], "data": "synthetic",
            "config": {"workload": f"synthetic {args.grid}x{rows}-node triangle mesh, one rollout step (predict) "
                                   f"= {args.mp_steps} MP steps + encoder + decoder + integrator",
                       "nodes": n, "edges": e, "mp_steps": args.mp_steps, "latent": 128, "node_order": "row-major" if args.reorder == "none" else args.reorder,
                       "parallelism": (f"node-range partition x{world_size} ({args.grid}x{rows // world_size} nodes per GPU, halo "
                                       "exchange of the sender projections per MP step)") if world_size > 1 else "single GPU"},
            "edge_updates_per_s_per_gpu": e * args.mp_steps * args.steps / dt / world_size,
            "rollout_steps_per_s": args.steps / dt,
            "finite": finite,
            "matmul_mode": mode,
            "ms_per_step_instrumented": dt_inst / args.steps * 1e3,
            "kernel_ms": kern_ms,
        }
        if strict is not None:
            out["strict_f32"] = strict
        if "mp_edge" in kern_ms:
            t_edge = kern_ms["mp_edge"] * 1e-3
            ach = FLOP_PER_EDGE_EDGE_KERNEL * e_kernel / t_edge / 1e12
            kernel, path, pipe_peak, products = {
                "f32": ("mp_edge_kernel", "v_mfma_f32_32x32x2_f32", PEAK_F32_MFMA_TFLOPS, 1),
                "bf16x3": ("mp_edge3_kernel", "fp32 operands split exactly into 3 bf16 terms, 6 partial products on "
                           "v_mfma_f32_32x32x16_bf16, fp32 accumulate", PEAK_BF16_MFMA_TFLOPS, 6),
            }.get(mode, ("mp_edge_h4_kernel", "fp32 operands split into 2 fp16 terms (residual 2^-22), 3 partial products on "
                         "v_mfma_f32_32x32x16_f16, fp32 accumulate", PEAK_F16_MFMA_TFLOPS, 3))
            traffic = pmc_traffic(kernel, e_kernel)          # counter pass of a launch that stores its edge latents
            if traffic is not None:
                traffic -= 512.0 / args.mp_steps * e_kernel  # same average over the launches as `per_edge` below
            # the last block of the stack does not store its edge latents: average over the launches timed
            per_edge = BYTES_PER_EDGE_EDGE_KERNEL - 512.0 / args.mp_steps
            bytes_alg = per_edge * e_kernel
            gbs = bytes_alg / t_edge / 1e9
            t_floor_mfma = products * FLOP_PER_EDGE_EDGE_KERNEL * e_kernel / (pipe_peak * 1e12)
            t_floor_hbm = bytes_alg / (PEAK_HBM_GBS * 1e9)
            # the roof that binds is the one with the larger floor for this kernel's executed work
            if t_floor_hbm >= t_floor_mfma:
                roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
            else:
                executed = products * ach
                roof = {"bound": "mfma", "achieved": executed, "peak": pipe_peak, "unit": "TFLOP/s", "frac": executed / pipe_peak}
            out["roofline"] = {"kernel": kernel, **roof, "traffic": traffic,
                               "traffic_unit": "bytes per launch (rocprofv3 PMC pass, see profiles/)",
                               "avg_launch_ms": kern_ms["mp_edge"],
                               "bytes_per_edge_algorithmic": per_edge,
                               "flop_per_edge_algorithmic": FLOP_PER_EDGE_EDGE_KERNEL,
                               "fp32_equivalent_TFLOPs": ach, "fp32_equivalent_frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                               "matrix_path": path, "matrix_products_per_fp32_product": products,
                               "matrix_pipe_frac": products * ach / pipe_peak,
                               "matrix_pipe_busy_measured": pmc_record().get("mfma_busy_frac") if pmc_record().get("kernel") == kernel else None,
                               "counters": {k: pmc_record().get(k) for k in ("wave_cycle_shares", "l2_hit_rate", "clock_GHz_under_profiler", "source")} if pmc_record().get("kernel") == kernel else None,
                               "floor_ms": {"hbm": t_floor_hbm * 1e3, "mfma": t_floor_mfma * 1e3},
                               # the counter traffic (e rows, node-side gathers and sums included) over this launch time,
                               # beside what plain streaming kernels reach on this GPU (scripts/micro/hbm_stream.hip, 3 GiB
                               # arrays: copy 5.0-5.4, 2 reads + 1 write 5.3-5.6 TB/s; profiles/r03_hbm_stream.txt) and the
                               # time of a kernel that only performs this kernel's memory traffic on the 1M-node mesh
                               # (scripts/micro/edge_stream.hip, profiles/r03_edge_stream.txt: 8 and 2 workgroups per CU)
                               "moved_GBs": (traffic / t_edge / 1e9) if traffic else None,
                               "streaming_kernel_GBs_measured": [5000.0, 5600.0],
                               "traffic_only_kernel_ms_measured_at_1M_nodes": [1.55, 1.69]}
            if "mp_node" in kern_ms:
                # second kernel of the step: node update + next block's projections, 2560 algorithmic bytes per node
                # (x and aggr in; x, ps, pr out); the last block writes no projections (averaged like the edge kernel)
                per_node = 2560.0 - 1024.0 / args.mp_steps
                t_node = kern_ms["mp_node"] * 1e-3
                out["roofline_node_kernel"] = {"kernel": "mp_node_h_kernel", "bound": "hbm", "achieved": per_node * n_local / t_node / 1e9,
                                               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": per_node * n_local / t_node / 1e9 / PEAK_HBM_GBS,
                                               "traffic": pmc_traffic("mp_node_h_kernel", n_local, ("r03_pmc_traffic_node.json",)),
                                               "traffic_unit": "bytes per launch that writes projections (rocprofv3 PMC passes, profiles/r03_pmc_traffic_node.json)",
                                               "avg_launch_ms": kern_ms["mp_node"], "bytes_per_node_algorithmic": per_node}
            # one MP step = everything the processor launches for one layer; the projection of
            # layers > 0 is fused into the previous layer's node kernel, so divide totals by layers
            n_layers = args.steps * args.mp_steps
            t_mp = sum(sum(per.get(k, [])) for k in ("mp_project", "halo_exchange", "mp_edge", "mp_edge_boundary", "mp_node")) * 1e-3 / n_layers
            canon = (FLOP_PER_EDGE_CANONICAL * e_local + FLOP_PER_NODE_CANONICAL * n_local) / t_mp / 1e12
            out["mp_step"] = {"ms": t_mp * 1e3, "edge_updates_per_s_per_gpu": e_local / t_mp,
                              "edge_updates_per_s": e / t_mp,
                              "canonical_TFLOPs": canon,
                              "compulsory_GBs": 1028.0 * (e_local + n_local) / t_mp / 1e9,
                              "compulsory_frac_of_hbm_peak": 1028.0 * (e_local + n_local) / t_mp / 1e9 / PEAK_HBM_GBS}
        if world_size == 1 and args.cpu_reps > 0:
            out["cpu_baseline"] = cpu_baseline(g_full, args.mp_steps, args.cpu_reps)
        if world_size == 1 and args.train_steps > 0:
            del model, g
            torch.cuda.empty_cache()
            # secondary objects: a failure in one of them is reported in its place and does not cost the headline line
            for key, fn in (("train", lambda: train_object(args, device)),
                            ("other_configs", lambda: other_configs(device, cpu=args.cpu_reps > 0))):
                try:
                    out[key] = fn()
                except Exception as ex:             # noqa: BLE001
                    out[key] = {"error": f"{type(ex).__name__}: {ex}"[:500]}
        print(json.dumps(out), flush=True)
    if world_size > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
