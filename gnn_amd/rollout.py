"""Autoregressive rollout with the state kept on the GPU.

Same call and same returned dictionary as ``core/rollout.py:5-59`` of the reference.
The reference moves the frame to the device on every step (rollout.py:28); here the
static fields and the whole swelling schedule are uploaded once and each step only
swaps device pointers.
"""
from __future__ import annotations

import os

import torch

from .graph import MeshGraph, batch_graphs, unbatch_nodes


GRAPH_AUTO_MAX_EDGES = 50_000


def _wants_graph(model, first, use_graph):
    """Graph replay: ``use_graph=True/False`` or ``MGN_ROLLOUT_GRAPH=1/0`` decide; left open, it is used for meshes
    of at most 50 000 edges.  There a step's ~40 launches occupy the GPU for less time (0.35 ms at 561 nodes) than
    the host needs to issue them (0.45 ms), and replaying the captured step takes 0.39 ms; at 200 000 edges the
    host keeps ahead of the GPU and replay is the slower of the two.  Results are bit-identical either way.  Not
    with a partitioned model, and with the coarse branch only while its first sample is pinned
    (``model.coarse_start``: otherwise the coarse graph is redrawn per call)."""
    if use_graph is None:
        env = os.environ.get("MGN_ROLLOUT_GRAPH", "")
        use_graph = first["edge_index"].shape[1] <= GRAPH_AUTO_MAX_EDGES if env == "" else env != "0"
    return (bool(use_graph) and model.__dict__.get("halo") is None and torch.device(model.device).type == "cuda"
            and (getattr(model, "sample_ratio", 0.0) == 0 or model.__dict__.get("coarse_start") is not None))


class _StepGraph:
    """One rollout step -- ``predict`` plus the state feedback -- captured as a HIP graph.

    Replaying the captured sequence is one launch instead of ~45; it pays where the host is the
    bottleneck (slow host cores, many small meshes driven from one thread).  What changes from step
    to step lives in fixed device buffers: the state (fed back inside the graph), the swelling
    window (copied in before each replay) and the normalisers' row counts
    (``Normalizer.use_device_count``)."""

    def __init__(self, model, cur):
        self.model, self.cur = model, cur
        self.plain = getattr(model, "sample_ratio", 0.0) == 0
        # the feature normalisers (they accumulate while live), then the output normaliser (read only)
        feature_norms = (model.node_features_normalizer, model.edge_features_normalizer)
        if not self.plain:
            feature_norms += (model.coarse_edge_features_normalizer,)
        self.n_feature_norms = len(feature_norms)
        self.norms = feature_norms + (model.output_normalizer,)
        for nr in self.norms:
            nr.use_device_count(True)
        self.accumulating = [nr._num_accumulations < nr._max_accumulations for nr in feature_norms]
        state = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):             # a real pass: creates every lazily made buffer outside the capture
            self._body()
        torch.cuda.current_stream().wait_stream(side)
        self._restore(state)
        counts = [(nr._acc_count, nr._num_accumulations) for nr in self.norms]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.pred = self._body()
        self.rows_added = [nr._acc_count - c for nr, (c, _) in zip(self.norms, counts)]
        for nr, (c, k) in zip(self.norms, counts):   # the capture ran the host bookkeeping once without executing
            nr._acc_count, nr._num_accumulations = c, k

    def _body(self):
        if self.plain:                            # the decoder launch writes the new state over the old one
            return self.model.predict(self.cur, check_range=False, feedback=True)
        pred = self.model.predict(self.cur, check_range=False)
        last = pred[-1]
        self.cur["world_pos"].copy_(last[:, :2])
        self.cur["phi"].copy_(last[:, 2:])
        return pred

    def _snapshot(self):
        norms = [(nr._acc_sum.clone(), nr._acc_sum_squared.clone(), nr._acc_count, nr._num_accumulations) for nr in self.norms]
        return norms, self.cur["world_pos"].clone(), self.cur["phi"].clone()

    def _restore(self, state):
        norms, world, phi = state
        for nr, (s1, s2, cnt, num) in zip(self.norms, norms):
            nr._acc_sum.copy_(s1); nr._acc_sum_squared.copy_(s2)
            nr._acc_count, nr._num_accumulations = cnt, num
            nr.use_device_count(True)               # device copy of the count back in step
        self.cur["world_pos"].copy_(world); self.cur["phi"].copy_(phi)

    def valid(self) -> bool:
        """The captured sequence accumulates (or not) as it did at capture time."""
        return self.accumulating == [nr._num_accumulations < nr._max_accumulations for nr in self.norms[: self.n_feature_norms]]

    def step(self):
        self.graph.replay()
        for nr, rows, acc in zip(self.norms[: self.n_feature_norms], self.rows_added, self.accumulating):
            if acc:
                nr.count_replayed(rows)
        return self.pred

    def close(self):
        for nr in self.norms:
            nr.use_device_count(False)


def rollout(model, data, progress: bool = False, reorder=None, use_graph=None):
    """``reorder``: None, or "rcm" / "morton" to renumber the nodes for locality before the
    first step (meshes whose ids do not follow the geometry); every returned tensor stays in
    the order of ``data``.  ``use_graph``: replay each step as one captured HIP graph (default: for small
    meshes, see :func:`_wants_graph`; ``MGN_ROLLOUT_GRAPH=0/1`` overrides)."""
    if reorder is not None:
        return _rollout_reordered(model, data, progress, reorder, use_graph)
    device = model.device
    time_dim = model.time_dim
    n_frames = len(data)
    first = data[0]
    import contextlib
    frozen = model.constant_weights() if hasattr(model, "constant_weights") else contextlib.nullcontext()
    with torch.no_grad(), frozen:
        static = MeshGraph(mesh_pos=first["mesh_pos"], node_type=first["node_type"], mat_param=first["mat_param"],
                           edge_index=first["edge_index"]).to(device)
        if "batch_sizes" in first:                # block-diagonal batch: the coarse branch samples every block on its own
            static["batch_sizes"] = first["batch_sizes"]
        # swelling schedule [T + time_dim, N], zero past the end (rollout.py:17-26)
        sched = torch.stack([data[t]["swelling_phi"] for t in range(n_frames)]
                            + [torch.zeros_like(first["swelling_phi"])] * time_dim).to(device)
        cur = MeshGraph(**dict(static.items()))
        cur["world_pos"] = first["world_pos"].to(device, torch.float32).clone()
        cur["phi"] = first["phi"].to(device, torch.float32).clone()
        preds = [torch.cat([cur["world_pos"], cur["phi"]], dim=-1).unsqueeze(0)]
        steps = range(0, n_frames, time_dim)
        if progress:
            from tqdm import tqdm
            steps = tqdm(steps, desc="Rollout")
        captured = None
        # plain model on one GPU: the decoder launch writes the new state over the old one (no separate copies)
        in_place = (getattr(model, "__dict__", {}).get("halo") is None and getattr(model, "sample_ratio", 0.0) == 0
                    and hasattr(model, "constant_weights"))
        if _wants_graph(model, first, use_graph) and n_frames > 2 * time_dim:
            cur["swelling_phi"] = sched[0: time_dim + 1].T.contiguous().to(torch.float32)
            captured = _StepGraph(model, cur)
        try:
            for t in steps:
                if captured is not None and captured.valid():
                    cur["swelling_phi"].copy_(sched[t: t + time_dim + 1].T)
                    preds.append(captured.step().clone())
                    continue
                cur["swelling_phi"] = sched[t: t + time_dim + 1].T.contiguous()
                if in_place:
                    pred = model.predict(cur, check_range=False, feedback=True)       # range checked once, after the loop
                else:
                    pred = model.predict(cur, check_range=False)
                    last = pred[-1]
                    cur["world_pos"] = last[:, :2].contiguous()
                    cur["phi"] = last[:, 2:].contiguous()
                preds.append(pred)
        finally:
            if captured is not None:
                captured.close()
        from . import ops
        ops.raise_if_out_of_range("rollout")
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


def _rollout_reordered(model, data, progress, method, use_graph=None):
    from . import reorder as R
    order = R.node_order(data[0], method)
    edges = R.permute_frame({"mesh_pos": data[0]["mesh_pos"], "edge_index": data[0]["edge_index"]}, order)["edge_index"]
    res = rollout(model, [R.permute_frame(fr, order, relabelled_edges=edges) for fr in data], progress, use_graph=use_graph)
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
    the per-trajectory results are identical to separate :func:`rollout` calls.  With the coarse multiscale branch every
    block is sampled and connected on its own (``model.coarse_start`` is the first sample inside each block), so no
    coarse edge ties two trajectories together."""
    trajectories = list(trajectories)
    n_frames = len(trajectories[0])
    if any(len(tr) != n_frames for tr in trajectories):
        raise ValueError("rollout_batched: trajectories must have the same number of frames")
    # frame 0 carries the static fields of the union; of the later frames the rollout reads only the
    # per-step fields, which are merged for all frames at once (one stack per trajectory and field)
    merged0 = batch_graphs([tr[0] for tr in trajectories])
    sizes = merged0["batch_sizes"]
    series = {k: torch.cat([torch.stack([fr[k] for fr in tr]) for tr in trajectories], dim=1)
              for k in ("world_pos", "phi", "swelling_phi")}
    merged = [merged0] + [MeshGraph(time=trajectories[0][t]["time"], **{k: v[t] for k, v in series.items()})
                          for t in range(1, n_frames)]
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
