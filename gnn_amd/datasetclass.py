"""Edge build and frame construction.

``cells_to_edge_index`` replaces core/datasetclass.py:7-20 of the reference (including
the PyG ``to_undirected`` call at :19) with the host routine of libmgn_hip.so; the
result is identical, integer for integer.  ``HydrogelDataset`` keeps the reference's
``.npz`` schema (core/datasetclass.py:43-55; gt_gen/gt_gen/bending_gen.py:297-306) and
returns ``MeshGraph`` frames instead of PyG ``Data``.
"""
from __future__ import annotations

import os
from typing import List

import numpy as np
import torch

from . import ops
from .graph import MeshGraph


def cells_to_edge_index(cells: torch.Tensor) -> torch.Tensor:
    """Triangular cells [C,3] -> directed, duplicate-free edge_index [2,E] (int64), sorted by
    (sender, receiver), both directions of every triangle side present."""
    if torch.is_tensor(cells) and cells.is_cuda:          # mesh already in HBM: build the edges there
        return ops.cells_to_edge_index_dev(cells.detach())[0]
    arr = cells.detach().cpu().numpy() if torch.is_tensor(cells) else np.asarray(cells)
    ei, _ = ops.cells_to_edge_index_np(arr)
    return torch.from_numpy(ei)


def grid_mesh(nx: int, ny: int, lx: float = 1.0, ly: float = 1.0):
    """(nx+1)*(ny+1) nodes on [0,lx]x[0,ly], ids row-major (x fastest); each cell split along
    the diagonal from its lower-left to its upper-right corner.  Returns (pos f32 [N,2], cells i64 [C,3])."""
    xs = np.linspace(0.0, lx, nx + 1)
    ys = np.linspace(0.0, ly, ny + 1)
    pos = np.empty(((nx + 1) * (ny + 1), 2), dtype=np.float32)
    pos[:, 0] = np.tile(xs, ny + 1)
    pos[:, 1] = np.repeat(ys, nx + 1)
    ll = (np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]).reshape(-1)
    cells = np.empty((2 * nx * ny, 3), dtype=np.int64)
    cells[0::2, 0], cells[0::2, 1], cells[0::2, 2] = ll, ll + 1, ll + nx + 2
    cells[1::2, 0], cells[1::2, 1], cells[1::2, 2] = ll, ll + nx + 1, ll + nx + 2
    return pos, cells


def synthetic_frames(nx: int, ny: int, n_frames: int = 1, seed: int = 0, lx: float = 1.0, ly: float = 1.0,
                     time_dim: int = 1) -> List[MeshGraph]:
    """Seeded synthetic trajectory on a structured mesh with bending-style boundary flags
    (left column: ux, uy fixed; top row: swelling boundary).  Frame fields follow the
    reference's frames without targets (core/datasetclass.py:110-118)."""
    pos, cells = grid_mesh(nx, ny, lx, ly)
    n = pos.shape[0]
    gen = torch.Generator().manual_seed(seed)
    mesh_pos = torch.from_numpy(pos)
    ids = torch.arange(n)
    left, top = (ids % (nx + 1)) == 0, (ids // (nx + 1)) == ny
    node_type = torch.zeros(n, 5)
    node_type[left, 1] = 1.0
    node_type[left, 2] = 1.0
    node_type[top, 4] = 1.0
    node_type[~(left | top), 0] = 1.0
    edge_index = cells_to_edge_index(torch.from_numpy(cells))
    cells_t = torch.from_numpy(cells)
    mat_param = torch.tensor([0.2, 7.5e-6])
    world0 = mesh_pos + 1e-3 * torch.randn(n, 2, generator=gen)
    phi0 = 0.5 + 0.1 * torch.rand(n, 1, generator=gen)
    frames = []
    for t in range(n_frames):
        frames.append(MeshGraph(mesh_pos=mesh_pos, node_type=node_type, mat_param=mat_param, cells=cells_t,
                                edge_index=edge_index, time=torch.tensor(float(t)),
                                world_pos=world0 + 1e-5 * t, phi=phi0 + 1e-4 * t,
                                swelling_phi=(0.35 + 0.002 * t) * node_type[:, 4]))
    if n_frames == 1:       # single-frame use: give the window the model expects, [N, time_dim + 1]
        frames[0]["swelling_phi"] = frames[0]["swelling_phi"].unsqueeze(1).repeat(1, time_dim + 1)
    return frames


def synthetic_strip(nx: int, ny: int, world_size: int, rank: int, seed: int = 0, lx: float = 1.0, ly: float = 1.0,
                    time_dim: int = 1):
    """One rank's part of a synthetic frame on the (nx+1) x (ny+1)-node grid mesh of :func:`grid_mesh`, built
    without forming the whole mesh: the mesh rows that hold the rank's node range (as
    :func:`gnn_amd.partition.node_bounds` cuts it) plus one row on either side.  Boundary flags as in
    :func:`synthetic_frames`; the random state of a node depends only on (seed, its mesh row, its column), so
    every rank -- and a single-rank run -- sees the same values for the same node.

    Returns ``(frame, in_edges, out_edges, num_nodes, num_edges)``: ``frame`` holds the strip's nodes in global
    order with ``node_ids`` (their global ids) and no ``edge_index``; ``in_edges`` / ``out_edges`` [2, .] are the
    edge sets :meth:`gnn_amd.partition.HaloPlan.from_local` takes, in global ids; the counts are global."""
    from .partition import node_bounds
    w = nx + 1
    n_total = w * (ny + 1)
    e_total = 2 * (3 * nx * ny + nx + ny)                       # both directions of every grid side and diagonal
    bounds = node_bounds(n_total, world_size)
    lo, hi = bounds[rank], bounds[rank + 1]
    ya, yb = max(lo // w - 1, 0), min((hi - 1) // w + 1, ny)      # first and last mesh row of the strip
    rows = yb - ya + 1
    g0 = ya * w
    n = rows * w
    xs = np.linspace(0.0, lx, nx + 1)
    ys = np.linspace(0.0, ly, ny + 1)[ya: yb + 1]
    pos = np.empty((n, 2), dtype=np.float32)
    pos[:, 0] = np.tile(xs, rows)
    pos[:, 1] = np.repeat(ys, w)
    ll = (np.arange(rows - 1)[:, None] * w + np.arange(nx)[None, :]).reshape(-1)
    cells = np.empty((2 * nx * (rows - 1), 3), dtype=np.int64)
    cells[0::2, 0], cells[0::2, 1], cells[0::2, 2] = ll, ll + 1, ll + nx + 2
    cells[1::2, 0], cells[1::2, 1], cells[1::2, 2] = ll, ll + nx + 1, ll + nx + 2
    ei = cells_to_edge_index(torch.from_numpy(cells)).numpy() + g0
    snd, rcv = ei[0], ei[1]
    mine = (rcv >= lo) & (rcv < hi)
    leaving = (snd >= lo) & (snd < hi) & ~mine
    ids = torch.arange(g0, g0 + n)
    left, top = (ids % w) == 0, (ids // w) == ny
    node_type = torch.zeros(n, 5)
    node_type[left, 1] = 1.0
    node_type[left, 2] = 1.0
    node_type[top, 4] = 1.0
    node_type[~(left | top), 0] = 1.0
    mesh_pos = torch.from_numpy(pos)
    noise = torch.empty(rows, w, 3)
    for k in range(rows):                                       # one small generator per mesh row
        gen = torch.Generator().manual_seed(seed * 1_000_003 + ya + k)
        noise[k, :, :2] = torch.randn(w, 2, generator=gen)
        noise[k, :, 2] = torch.rand(w, generator=gen)
    noise = noise.reshape(n, 3)
    frame = MeshGraph(mesh_pos=mesh_pos, node_type=node_type, mat_param=torch.tensor([0.2, 7.5e-6]),
                      time=torch.tensor(0.0), world_pos=mesh_pos + 1e-3 * noise[:, :2],
                      phi=0.5 + 0.1 * noise[:, 2:], node_ids=ids,
                      swelling_phi=(0.35 * node_type[:, 4]).unsqueeze(1).repeat(1, time_dim + 1))
    return frame, ei[:, mine], ei[:, leaving], n_total, e_total


def strip_local_graph(frame: MeshGraph, plan) -> MeshGraph:
    """The [owned | halo] rows of a :func:`synthetic_strip` frame for ``plan``, ``edge_index`` in local ids."""
    idx = torch.from_numpy(plan.local_node_ids) - int(frame["node_ids"][0])
    out = MeshGraph()
    n = frame["mesh_pos"].shape[0]
    for k, v in frame.items():
        if k == "node_ids":
            continue
        out[k] = v[idx] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n else v
    out["edge_index"] = torch.from_numpy(plan.local_edge_index)
    return out


class DeviceTrajectory:
    """One trajectory resident on the GPU; frames are views of it.

    The reference builds T python frames per ``__getitem__`` on the host and the training loop
    moves each of them to the device (core/datasetclass.py:43-108, run_train_multiscale.py:137-138).
    Here the ``[T, N, .]`` series are uploaded once; ``frame(t)`` returns the fields of the
    reference's frame t (:97-106) as device tensors -- target and state are slices of the resident
    series, the noise (:75-89) is injected by ``mgn_inject_noise`` from device-side normal draws --
    so a training step does no host-to-device copy.  ``edge_index`` is one tensor for all frames,
    hence one cached topology."""

    def __init__(self, sample, device, time_dim: int = 1):
        """``sample``: mapping with the ``.npz`` schema of the reference's generators
        (bending_gen.py:297-306) -- a loaded ``.npz`` or a dict of arrays."""
        f32 = lambda k: torch.as_tensor(np.asarray(sample[k]), dtype=torch.float32)
        self.device, self.time_dim = torch.device(device), int(time_dim)
        mesh_pos = f32("mesh_coords")
        cells = torch.as_tensor(np.asarray(sample["cells"]), dtype=torch.long)
        edge_index = cells_to_edge_index(cells)
        node_type = f32("node_type")
        world = mesh_pos + f32("u_time_series")
        phi = f32("φ_time_series").unsqueeze(-1)
        swell = f32("swell_time_series").reshape(-1, 1)
        swell_nodes = node_type[:, 4] == 1
        swell_full = torch.zeros(phi.shape[0], phi.shape[1])
        swell_full[:, swell_nodes] = swell.expand(phi.shape[0], int(swell_nodes.sum()))
        row, col = edge_index
        # scales of the noise model, constants of the trajectory (core/datasetclass.py:76, :83)
        self.conn_length = float(torch.max(torch.norm(mesh_pos[row] - mesh_pos[col], dim=-1))) if edge_index.numel() else 0.0
        self.phi_range = float(torch.max(phi) - torch.min(phi))
        dev = self.device
        self.mesh_pos, self.node_type, self.cells = mesh_pos.to(dev), node_type.to(dev), cells.to(dev)
        self.edge_index = edge_index.to(dev)
        self.mat_param = torch.tensor([np.asarray(sample["chi"]).item(), np.asarray(sample["diffusivity"]).item()],
                                      dtype=torch.float32, device=dev)
        self.time = f32("t").to(dev)
        self.world_pos, self.phi = world.to(dev), phi.to(dev)                  # [T, N, 2], [T, N, 1]
        self.state = torch.cat([self.world_pos, self.phi], dim=-1)             # [T, N, 3]: targets are slices of it
        self.swelling = swell_full.T.contiguous().to(dev)                      # [N, T]
        self.num_steps = world.shape[0]

    def __len__(self):
        """Number of training frames (those with ``time_dim`` successors, core/datasetclass.py:66-72)."""
        return self.num_steps - self.time_dim

    def frame(self, t: int, noise_level: float = 0.0, generator=None, with_target: bool = True) -> MeshGraph:
        td = self.time_dim
        if not 0 <= t < (len(self) if with_target else self.num_steps):
            raise IndexError(f"frame {t} out of range")
        world, phi = self.world_pos[t], self.phi[t]
        if noise_level and noise_level > 0.0:
            n = world.shape[0]
            zw = torch.randn(n, 2, device=self.device, generator=generator)
            zp = torch.randn(n, 1, device=self.device, generator=generator)
            nw, nphi = torch.empty_like(world), torch.empty_like(phi)
            ops.inject_noise(world, phi, self.node_type, zw, zp, noise_level, self.conn_length, self.phi_range, nw, nphi)
            world, phi = nw, nphi
        g = MeshGraph(mesh_pos=self.mesh_pos, node_type=self.node_type, mat_param=self.mat_param, cells=self.cells,
                      edge_index=self.edge_index, time=self.time[t], world_pos=world, phi=phi)
        if with_target:
            g["swelling_phi"] = self.swelling[:, t: t + td + 1].contiguous()    # [N, time_dim + 1]
            g["target"] = self.state[t + 1: t + 1 + td]                         # [time_dim, N, 3]
        else:
            g["swelling_phi"] = self.swelling[:, t].contiguous()
        return g

    def frames(self, noise_level: float = 0.0, generator=None):
        return [self.frame(t, noise_level, generator) for t in range(len(self))]


class HydrogelDataset(torch.utils.data.Dataset):
    """``.npz`` trajectories -> lists of frames (core/datasetclass.py:22-119): same constructor
    arguments, same fields per frame, same noise model (:75-89) and multi-step targets (:66-72)."""

    def __init__(self, data_dir, noise_level=None, add_targets=None, split_to_frames=None, time_dim=5):
        self.data_dir = data_dir
        self.noise_level = noise_level
        self.add_targets = add_targets
        self.split_to_frames = split_to_frames
        self.time_dim = time_dim
        self.data_files = [f for f in os.listdir(data_dir) if f.endswith(".npz")]

    def __len__(self):
        return len(self.data_files)

    def get_name(self, idx):
        return self.data_files[idx].rstrip(".npz")

    def device_trajectory(self, idx, device) -> DeviceTrajectory:
        """Trajectory ``idx`` uploaded once; its ``frame(t, noise_level)`` replaces the host frames."""
        return DeviceTrajectory(np.load(os.path.join(self.data_dir, self.data_files[idx])), device, self.time_dim)

    def __getitem__(self, idx):
        s = np.load(os.path.join(self.data_dir, self.data_files[idx]))
        f32 = lambda k: torch.tensor(s[k], dtype=torch.float32)
        mesh_pos = f32("mesh_coords")
        cells = torch.tensor(s["cells"], dtype=torch.long)
        edge_index = cells_to_edge_index(cells)
        node_type = f32("node_type")
        world_pos = mesh_pos + f32("u_time_series")
        phi = f32("φ_time_series").unsqueeze(-1)
        swell = f32("swell_time_series").unsqueeze(-1)
        time = f32("t")
        mat_param = torch.tensor([s["chi"].item(), s["diffusivity"].item()], dtype=torch.float32)
        swell_nodes = node_type[:, 4] == 1
        swell_full = torch.zeros(phi.shape[0], phi.shape[1])
        swell_full[:, swell_nodes] = swell.expand(phi.shape[0], int(swell_nodes.sum()))
        common = dict(mesh_pos=mesh_pos, node_type=node_type, mat_param=mat_param, cells=cells, edge_index=edge_index)
        if not self.add_targets:
            return [MeshGraph(**common, time=time[t], world_pos=world_pos[t], phi=phi[t], swelling_phi=swell_full[t])
                    for t in range(world_pos.shape[0])]
        td = self.time_dim
        n_cur = world_pos.shape[0] - td
        row, col = edge_index
        frames = []
        for t in range(n_cur):
            wp, ph = world_pos[t], phi[t]
            if self.noise_level and self.noise_level > 0.0:
                conn = torch.max(torch.norm(mesh_pos[row] - mesh_pos[col], dim=-1))
                wn = torch.randn_like(wp) * self.noise_level * conn
                wn[node_type[:, 1] == 1, 0] = 0.0
                wn[node_type[:, 2] == 1, 1] = 0.0
                pn = torch.randn_like(ph) * self.noise_level * (torch.max(phi) - torch.min(phi))
                pn[node_type[:, 3] == 1] = 0.0
                wp, ph = wp + wn, ph + pn
            target = torch.cat([world_pos[t + 1: t + 1 + td], phi[t + 1: t + 1 + td]], dim=-1)
            frames.append(MeshGraph(**common, time=time[t], world_pos=wp, phi=ph,
                                    swelling_phi=swell_full[t: t + td + 1].T, target=target))
        return frames
