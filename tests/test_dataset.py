"""Frames from ``.npz`` trajectories: the host dataset and the resident trajectory against frames the
reference's own ``HydrogelDataset`` produced (fixture g6_dataset.npz, written by tests/golden/make_golden.py
from core/datasetclass.py:22-119 run in the build container), against the reference's frame layout
restated field by field, and the resident trajectory against the host dataset.  The GPU part checks the device noise kernel bit for bit against the oracle's
restatement of :75-89 with the same normal draws, and that a training step runs from a resident
frame."""
import numpy as np
import pytest
import torch

import oracle as O


def _write_npz(path, nx=6, ny=4, T=9, seed=0):
    rng = np.random.default_rng(seed)
    pos, cells = O.grid_mesh(nx, ny)
    n = pos.shape[0]
    node_type = O.bending_node_type(nx, ny).astype(np.float32)
    np.savez(path, mesh_coords=pos.astype(np.float32), cells=cells, node_type=node_type,
             u_time_series=(1e-2 * rng.standard_normal((T, n, 2))).astype(np.float32),
             **{"φ_time_series": (0.5 + 0.1 * rng.random((T, n))).astype(np.float32)},
             swell_time_series=np.linspace(0.3, 0.4, T).astype(np.float32), t=np.arange(T, dtype=np.float32),
             chi=np.float32(0.2), diffusivity=np.float32(7.5e-6))
    return n


@pytest.mark.parametrize("time_dim", [1, 3])
def test_host_frames_follow_reference_layout(tmp_path, time_dim):
    from gnn_amd.datasetclass import HydrogelDataset
    n = _write_npz(tmp_path / "a.npz")
    s = np.load(tmp_path / "a.npz")
    ds = HydrogelDataset(str(tmp_path), noise_level=0.0, add_targets=True, time_dim=time_dim)
    assert len(ds) == 1 and ds.get_name(0) == "a"
    frames = ds[0]
    T = s["t"].shape[0]
    assert len(frames) == T - time_dim
    world = torch.tensor(s["mesh_coords"]) + torch.tensor(s["u_time_series"])
    phi = torch.tensor(s["φ_time_series"]).unsqueeze(-1)
    top = torch.tensor(s["node_type"])[:, 4] == 1
    for t in (0, 2, T - time_dim - 1):
        fr = frames[t]
        assert torch.equal(fr["world_pos"], world[t]) and torch.equal(fr["phi"], phi[t])
        assert fr["swelling_phi"].shape == (n, time_dim + 1) and fr["target"].shape == (time_dim, n, 3)
        assert torch.equal(fr["target"], torch.cat([world[t + 1: t + 1 + time_dim], phi[t + 1: t + 1 + time_dim]], -1))
        for k in range(time_dim + 1):
            exp = torch.where(top, torch.tensor(s["swell_time_series"][t + k]), torch.tensor(0.0))
            assert torch.equal(fr["swelling_phi"][:, k], exp)
        assert torch.equal(fr["edge_index"], torch.from_numpy(O.cells_to_edge_index(s["cells"])))
    plain = HydrogelDataset(str(tmp_path), add_targets=False, time_dim=time_dim)[0]
    assert len(plain) == T and plain[3]["swelling_phi"].shape == (n,) and "target" not in plain[3]


@pytest.mark.parametrize("time_dim", [1, 3])
def test_resident_trajectory_equals_host_frames(tmp_path, time_dim):
    from gnn_amd.datasetclass import HydrogelDataset
    _write_npz(tmp_path / "a.npz", seed=1)
    ds = HydrogelDataset(str(tmp_path), noise_level=0.0, add_targets=True, time_dim=time_dim)
    host = ds[0]
    traj = ds.device_trajectory(0, "cpu")
    assert len(traj) == len(host)
    for t in range(len(host)):
        fr = traj.frame(t)
        assert set(fr.keys()) == set(host[t].keys())
        for k in host[t].keys():
            assert torch.equal(fr[k], host[t][k]), (t, k)
    assert traj.frame(0)["edge_index"] is traj.frame(1)["edge_index"]        # one topology for all frames
    no_target = traj.frame(len(host) + time_dim - 1, with_target=False)
    assert "target" not in no_target and no_target["swelling_phi"].dim() == 1
    with pytest.raises(IndexError):
        traj.frame(len(host))


@pytest.mark.gpu
def test_device_noise_matches_reference_formula(tmp_path):
    from gnn_amd import ops
    from gnn_amd.datasetclass import HydrogelDataset
    n = _write_npz(tmp_path / "a.npz", nx=40, ny=30, seed=2)
    ds = HydrogelDataset(str(tmp_path), noise_level=0.02, add_targets=True, time_dim=2)
    traj = ds.device_trajectory(0, "cuda:0")
    gen = torch.Generator(device="cuda:0").manual_seed(7)
    zw, zp = torch.randn(n, 2, device="cuda:0", generator=gen), torch.randn(n, 1, device="cuda:0", generator=gen)
    fr = traj.frame(3, noise_level=0.02, generator=torch.Generator(device="cuda:0").manual_seed(7))
    ew, ep = O.inject_noise(traj.world_pos[3].cpu(), traj.phi[3].cpu(), traj.node_type.cpu(), zw.cpu(), zp.cpu(), 0.02,
                            torch.tensor(traj.conn_length), torch.tensor(traj.phi_range))
    assert torch.equal(fr["world_pos"].cpu(), ew) and torch.equal(fr["phi"].cpu(), ep)       # bit-exact
    nt = traj.node_type.cpu()
    assert torch.equal(fr["world_pos"].cpu()[nt[:, 1] == 1, 0], traj.world_pos[3].cpu()[nt[:, 1] == 1, 0])
    assert not torch.equal(fr["world_pos"], traj.world_pos[3])
    # the scales are the reference's (core/datasetclass.py:76, :83)
    s = np.load(tmp_path / "a.npz")
    ei = O.cells_to_edge_index(s["cells"])
    mp = torch.tensor(s["mesh_coords"])
    assert traj.conn_length == float(torch.max(torch.norm(mp[ei[0]] - mp[ei[1]], dim=-1)))
    assert traj.phi_range == float(torch.tensor(s["φ_time_series"]).max() - torch.tensor(s["φ_time_series"]).min())


@pytest.mark.gpu
def test_training_step_from_resident_frame(tmp_path):
    from gnn_amd.datasetclass import HydrogelDataset
    from test_gpu_model import build_model
    _write_npz(tmp_path / "a.npz", nx=12, ny=9, seed=3)
    cfg = O.OracleConfig(process_steps=2, time_dim=2, node_in_dim=11)
    sd = O.init_state_dict(cfg, 1)
    ds = HydrogelDataset(str(tmp_path), noise_level=0.0, add_targets=True, time_dim=2)
    a, b = build_model(cfg, sd), build_model(cfg, sd)
    la = a.loss(ds.device_trajectory(0, "cuda:0").frame(1))[0]
    lb = b.loss(ds[0][1].to("cuda:0"))[0]
    la.backward(); lb.backward()
    assert torch.isfinite(la) and torch.equal(la, lb)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p.grad, q.grad), k


# --------------------------------------------------------------------------- against the reference's own frames (G6)
G6_CONFIGS = [(True, 1, 0.0), (True, 1, 0.02), (True, 2, 0.0), (True, 2, 0.02), (False, 1, 0.0), (False, 2, 0.0)]


def _g6(tmp_path):
    from conftest import GOLDEN
    d = np.load(f"{GOLDEN}/g6_dataset.npz")
    np.savez(tmp_path / "traj_a.npz", **{k[len("npz_"):]: d[k] for k in d.files if k.startswith("npz_")})
    return d


def _tag(add_targets, time_dim, noise):
    return f"tgt{int(add_targets)}_td{time_dim}_noise{int(noise > 0)}"


@pytest.mark.parametrize("add_targets,time_dim,noise", G6_CONFIGS)
def test_host_dataset_reproduces_reference_frames_bit_for_bit(tmp_path, add_targets, time_dim, noise):
    """Same .npz, same seed: every field of every frame equals what core/datasetclass.py:43-119 returned, the
    Gaussian noise of :75-89 included (same draws from torch's global generator, same arithmetic)."""
    from gnn_amd.datasetclass import HydrogelDataset
    d = _g6(tmp_path)
    ds = HydrogelDataset(str(tmp_path), noise_level=noise, add_targets=add_targets, time_dim=time_dim)
    assert len(ds) == 1 and ds.get_name(0) == "traj_a"
    torch.manual_seed(123)
    frames = ds[0]
    tag = _tag(add_targets, time_dim, noise)
    assert len(frames) == int(d[tag + "_n_frames"])
    fields = ("time", "world_pos", "phi", "swelling_phi") + (("target",) if add_targets else ())
    for k in fields:
        got = torch.stack([fr[k] for fr in frames])
        ref = torch.from_numpy(d[f"{tag}_{k}"])
        assert got.dtype == ref.dtype and torch.equal(got, ref), k
    assert ("target" in frames[0]) == add_targets
    for k in ("mesh_pos", "node_type", "mat_param", "cells", "edge_index"):
        ref = torch.from_numpy(d["static_" + k])
        assert frames[0][k].dtype == ref.dtype and torch.equal(frames[0][k], ref), k
        assert all(fr[k] is frames[0][k] for fr in frames)


@pytest.mark.parametrize("time_dim", [1, 2])
def test_resident_trajectory_reproduces_reference_frames(tmp_path, time_dim):
    from gnn_amd.datasetclass import HydrogelDataset
    d = _g6(tmp_path)
    traj = HydrogelDataset(str(tmp_path), noise_level=0.0, add_targets=True, time_dim=time_dim).device_trajectory(0, "cpu")
    tag = _tag(True, time_dim, 0.0)
    assert len(traj) == int(d[tag + "_n_frames"])
    for t in range(len(traj)):
        fr = traj.frame(t)
        for k in ("time", "world_pos", "phi", "swelling_phi", "target"):
            assert torch.equal(fr[k], torch.as_tensor(d[f"{tag}_{k}"][t])), (t, k)
    plain = _tag(False, time_dim, 0.0)
    for t in (0, int(d[plain + "_n_frames"]) - 1):
        fr = traj.frame(t, with_target=False)
        for k in ("time", "world_pos", "phi", "swelling_phi"):
            assert torch.equal(fr[k], torch.as_tensor(d[f"{plain}_{k}"][t])), (t, k)


@pytest.mark.gpu
@pytest.mark.parametrize("time_dim", [1, 2])
def test_device_noise_kernel_reproduces_reference_noisy_frames(tmp_path, time_dim):
    """mgn_inject_noise fed the reference's own normal draws gives the reference's noisy frames bit for bit."""
    from gnn_amd import ops
    from gnn_amd.datasetclass import HydrogelDataset
    d = _g6(tmp_path)
    traj = HydrogelDataset(str(tmp_path), noise_level=0.02, add_targets=True, time_dim=time_dim).device_trajectory(0, "cuda:0")
    tag = _tag(True, time_dim, 0.02)
    level = float(d[tag + "_noise_level"])
    for t in range(int(d[tag + "_n_frames"])):
        zw, zp = torch.from_numpy(d[tag + "_z_world"][t]).cuda(), torch.from_numpy(d[tag + "_z_phi"][t]).cuda()
        w, p = torch.empty_like(traj.world_pos[t]), torch.empty_like(traj.phi[t])
        ops.inject_noise(traj.world_pos[t], traj.phi[t], traj.node_type, zw, zp, level, traj.conn_length, traj.phi_range, w, p)
        assert torch.equal(w.cpu(), torch.from_numpy(d[f"{tag}_world_pos"][t])), t
        assert torch.equal(p.cpu(), torch.from_numpy(d[f"{tag}_phi"][t])), t
