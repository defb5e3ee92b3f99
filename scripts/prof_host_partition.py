#!/usr/bin/env python3
"""Diagnostic (GPU box, one GPU): host issue time against GPU time of the PARTITIONED rollout step.

Runs rank R of W (default 3 of 8) of the 1000 x 1000-node mesh -- its own strip, its real plan, the real pack /
unpack kernels and launch sequence.  The peers do not exist on a one-GPU box, so the transport is a loop-back of
equal volume: `--transport nccl` issues the exchanges as real RCCL calls on a one-rank communicator (the host cost of
a collective call is what is being measured), `--transport copy` replaces them by a device copy.

Prints, per rollout step and per message-passing step: the wall time the host needs to issue the work on an idle GPU
(no synchronisation inside), and the time until the GPU has finished it.  The host keeps ahead of the GPU when the
first is smaller than the second.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_amd  # noqa: E402
from gnn_amd import partition  # noqa: E402
from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--grid", type=int, default=1000)
ap.add_argument("--mp-steps", type=int, default=15)
ap.add_argument("--transport", choices=("nccl", "copy"), default="nccl")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--profile", action="store_true", help="cProfile of the issuing thread")
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
group = None
if args.transport == "nccl":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


class LoopbackPlan(partition.HaloPlan):
    """Same volume, no peers: what is sent comes back (the values are meaningless, the work is not)."""

    def _all_to_all(self, recv, send, recv_counts, send_counts, async_op):
        rows = min(recv.shape[0], send.shape[0])
        if args.transport == "copy":
            recv[:rows].copy_(send[:rows])
            return None
        import torch.distributed as dist
        return dist.all_to_all_single(recv[:rows], send[:rows], output_split_sizes=[rows], input_split_sizes=[rows],
                                      async_op=async_op)

    def all_reduce_(self, t):
        if args.transport == "nccl":
            import torch.distributed as dist
            dist.all_reduce(t)
        return t


side = args.grid
frame, in_e, out_e, n, e = gnn_amd.synthetic_strip(side - 1, side - 1, args.world, args.rank, seed=0)
plan = LoopbackPlan.from_local(in_e, out_e, n, e, args.world, args.rank).to(dev)
torch.manual_seed(0)
model = EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=args.mp_steps, coarse_process_steps=0,
                                      node_out_dim=3, attention=False, with_mat_params=False, sample_ratio=0.0,
                                      time_dim=1, device=dev).to(dev).eval()
partition.attach(model, plan)
g = gnn_amd.strip_local_graph(frame, plan).to(dev)
# statistics frozen: the looped-back halo values are not a mesh state, and the timing should not depend on them
for nr in (model.node_features_normalizer, model.edge_features_normalizer):
    nr._acc_sum.fill_(0.0); nr._acc_sum_squared.fill_(1.0); nr._acc_count = 1


def step():
    pred = model.predict(g, check_range=False)
    last = pred[-1]
    g["world_pos"], g["phi"] = last[:, :2].contiguous(), last[:, 2:].contiguous()


for _ in range(5):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(args.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
host.sort(); total.sort()
h, t = host[len(host) // 2] * 1e3, total[len(total) // 2] * 1e3
print(f"rank {args.rank} of {args.world}, {side}x{side} mesh: owned {plan.num_owned} nodes, halo {plan.num_halo}, sends {plan.num_send} rows, "
      f"{plan.topology.num_edges} edges (interior {plan.interior[2]}, boundary {sum(r[2] for r in plan.boundary)}); transport {args.transport}")
print(f"per rollout step : host issue {h:.3f} ms, until the GPU is done {t:.3f} ms")
print(f"per MP step      : host issue {h / args.mp_steps:.4f} ms, until the GPU is done {t / args.mp_steps:.4f} ms "
      "(encoders, decoder and features included in both)")
if args.profile:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.reps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
if args.transport == "nccl":
    import torch.distributed as dist
    dist.destroy_process_group()
