"""Autoregressive rollout with the state kept on the GPU.

Same call and same returned dictionary as ``core/rollout.py:5-59`` of the reference.
The reference moves the frame to the device on every step (rollout.py:28); here the
static fields and the whole swelling schedule are uploaded once and each step only
swaps device pointers.
"""
from __future__ import annotations

import torch

from .graph import MeshGraph, batch_graphs, unbatch_nodes


def rollout(model, data, progress: bool = False, reorder=None):
    """``reorder``: None, or "rcm" / "morton" to renumber the nodes for locality before the
    first step (meshes whose ids do not follow the geometry); every returned tensor stays in
    the order of ``data``."""
    if reorder is not None:
        return _rollout_reordered(model, data, progress, reorder)
    device = model.device
    time_dim = model.time_dim
    n_frames = len(data)
    first = data[0]
    with torch.no_grad():
        static = MeshGraph(mesh_pos=first["mesh_pos"], node_type=first["node_type"], mat_param=first["mat_param"],
                           edge_index=first["edge_index"]).to(device)
        # swelling schedule [T + time_dim, N], zero past the end (rollout.py:17-26)
        sched = torch.stack([data[t]["swelling_phi"] for t in range(n_frames)]
                            + [torch.zeros_like(first["swelling_phi"])] * time_dim).to(device)
        cur = MeshGraph(**dict(static.items()))
        cur["world_pos"] = first["world_pos"].to(device).clone()
        cur["phi"] = first["phi"].to(device).clone()
        preds = [torch.cat([cur["world_pos"], cur["phi"]], dim=-1).unsqueeze(0)]
        steps = range(0, n_frames, time_dim)
        if progress:
            from tqdm import tqdm
            steps = tqdm(steps, desc="Rollout")
        for t in steps:
            cur["swelling_phi"] = sched[t: t + time_dim + 1].T.contiguous()
            pred = model.predict(cur)
            last = pred[-1]
            cur["world_pos"] = last[:, :2].contiguous()
            cur["phi"] = last[:, 2:].contiguous()
            preds.append(pred)
        rollout_preds = torch.cat(preds, dim=0)[:n_frames]
        rollout_gts = torch.stack([torch.cat([data[t]["world_pos"], data[t]["phi"]], dim=-1)
                                   for t in range(n_frames)]).to(device)
        swelling_phi = torch.stack([data[t]["swelling_phi"] for t in range(n_frames)])
        error = (rollout_preds - rollout_gts) ** 2
        rmse_x = torch.sqrt(torch.mean(error[:, :, 0]))
        rmse_y = torch.sqrt(torch.mean(error[:, :, 1]))
        rmse_phi = torch.sqrt(torch.mean(error[:, :, 2]))
    return {
        "time": torch.tensor([float(data[t]["time"]) for t in range(n_frames)]),
        "pred": rollout_preds,
        "gt": rollout_gts,
        "swelling_phi": swelling_phi,
        "mat_param": first["mat_param"],
        "mesh_pos": first["mesh_pos"],
        "cells": first["cells"],
        "node_type": first["node_type"],
        "rmse_x": rmse_x,
        "rmse_y": rmse_y,
        "rmse_phi": rmse_phi,
    }


def _rollout_reordered(model, data, progress, method):
    from . import reorder as R
    order = R.node_order(data[0], method)
    edges = R.permute_frame({"mesh_pos": data[0]["mesh_pos"], "edge_index": data[0]["edge_index"]}, order)["edge_index"]
    res = rollout(model, [R.permute_frame(fr, order, relabelled_edges=edges) for fr in data], progress)
    for k in ("pred", "gt", "swelling_phi"):
        res[k] = R.restore_nodes(res[k], order, dim=1)
    first = data[0]
    res.update(mesh_pos=first["mesh_pos"], cells=first["cells"], node_type=first["node_type"])
    return res


def rollout_batched(model, trajectories, progress: bool = False):
    """Advances several trajectories of equal length at once as one block-diagonal graph (the
    way small meshes fill the GPU) and returns one result dictionary per trajectory, each with
    the layout of :func:`rollout`.  The normalisers see one accumulation per step over all
    blocks instead of one per trajectory and step; with statistics that no longer accumulate
    the per-trajectory results are identical to separate :func:`rollout` calls."""
    trajectories = list(trajectories)
    n_frames = len(trajectories[0])
    if any(len(tr) != n_frames for tr in trajectories):
        raise ValueError("rollout_batched: trajectories must have the same number of frames")
    merged = [batch_graphs([tr[t] for tr in trajectories]) for t in range(n_frames)]
    sizes = merged[0]["batch_sizes"]
    res = rollout(model, merged, progress)
    preds, gts = unbatch_nodes(res["pred"], sizes, dim=1), unbatch_nodes(res["gt"], sizes, dim=1)
    swell = unbatch_nodes(res["swelling_phi"], sizes, dim=1)
    out = []
    for b, tr in enumerate(trajectories):
        err = (preds[b] - gts[b]) ** 2
        first = tr[0]
        out.append({
            "time": res["time"], "pred": preds[b], "gt": gts[b], "swelling_phi": swell[b],
            "mat_param": first["mat_param"], "mesh_pos": first["mesh_pos"], "cells": first["cells"],
            "node_type": first["node_type"],
            "rmse_x": torch.sqrt(torch.mean(err[:, :, 0])), "rmse_y": torch.sqrt(torch.mean(err[:, :, 1])),
            "rmse_phi": torch.sqrt(torch.mean(err[:, :, 2])),
        })
    return out
