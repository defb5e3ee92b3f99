"""One rank of the two-process GPU tests (started under torch.distributed.run by test_gpu_dist.py; not a
test module itself).  Backend from MGN_DIST_BACKEND: "nccl" = RCCL, one GPU per rank; "gloo" = both ranks on
GPU 0 with the halo rows staged through host memory.  Every rank builds only its own strip of the mesh, runs
three rollout steps and one training step through torch.distributed; rank 0 also runs the whole mesh alone and
prints the comparison as one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import gnn_amd
    import oracle as O
    from gnn_amd import partition, training
    from test_gpu_model import build_model
    backend = os.environ.get("MGN_DIST_BACKEND", "nccl")
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    idx = local if backend == "nccl" else 0
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    nx, ny, steps = 60, 45, 3
    cfg = O.OracleConfig(process_steps=4)
    sd = O.init_state_dict(cfg, 3)

    def run_steps(model, state):
        outs = []
        for _ in range(steps):
            pred = model.predict(state)
            last = pred[-1]
            state["world_pos"], state["phi"] = last[:, :2].contiguous(), last[:, 2:].contiguous()
            outs.append(pred)
        return torch.cat(outs)

    def loss_and_grads(model, graph):
        tgt = model._normalized_target(graph)
        loss3, saved = training.forward_saving(model, graph, tgt)
        grads = training.backward(model, saved, torch.ones(3, device=dev))
        return loss3.clone(), {k: v.clone() for k, v in grads.items()}

    def with_target(frame):
        n = frame["mesh_pos"].shape[0]
        ids = frame["node_ids"].double()                    # a target that depends on the global node id only
        bump = torch.stack([torch.sin(ids), torch.cos(ids), torch.sin(0.5 * ids)], -1).float() * 0.01
        frame["target"] = (torch.cat([frame["world_pos"], frame["phi"]], -1) + bump).unsqueeze(0)
        return frame

    frame, in_e, out_e, n, e = gnn_amd.synthetic_strip(nx, ny, world, rank, seed=5)
    frame = with_target(frame)
    plan = partition.HaloPlan.from_local(in_e, out_e, n, e, world, rank).to(dev)
    gl = gnn_amd.strip_local_graph(frame, plan)
    gl["target"] = frame["target"][:, torch.from_numpy(plan.local_node_ids) - int(frame["node_ids"][0])]
    gl = gl.to(dev)
    model = partition.attach(build_model(cfg, sd, dev), plan)
    loss, grads = loss_and_grads(model, gl)
    out = run_steps(model, gl)                              # [steps, owned, 3]
    full = plan.gather_owned(out.permute(1, 0, 2).contiguous()).permute(1, 0, 2)
    torch.cuda.synchronize()
    if rank == 0:
        whole, all_e, _, _, _ = gnn_amd.synthetic_strip(nx, ny, 1, 0, seed=5)
        whole = with_target(whole)
        g1 = gnn_amd.MeshGraph(**{k: v for k, v in whole.items() if k != "node_ids"})
        g1["edge_index"] = torch.from_numpy(all_e)
        g1 = g1.to(dev)
        ref_model = build_model(cfg, sd, dev)
        ref_loss, ref_grads = loss_and_grads(ref_model, g1)
        ref = run_steps(ref_model, g1)
        rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        print("DIST_RESULT " + json.dumps({
            "backend": backend, "world": world, "rollout_rel": rel(full, ref), "loss_rel": rel(loss, ref_loss),
            "grad_rel_worst": max(rel(grads[k], ref_grads[k]) for k in ref_grads),
            "finite": bool(torch.isfinite(full).all())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
