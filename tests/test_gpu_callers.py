"""GPU: the reference's two driver loops walked call by call with this package's classes in place of
core.* / torch_geometric.*: the training loop of run_train_multiscale.py:86-142 (dataset -> batch-size-1 shuffled
loader -> .to(device) -> loss -> backward -> Adam step -> .item()) and the rollout driver of
run_rollout_multiscale.py:46-84 (model from a selected_model/-layout directory -> dataset -> rollout -> the keys it
saves).  The CPU oracle walks the same sequences with the same seeds; losses and predictions must agree."""
import os

import numpy as np
import pytest
import torch

import oracle as O
from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


def _write_trajectory(tmp_path):
    d = np.load(f"{GOLDEN}/g6_dataset.npz")
    os.makedirs(tmp_path / "data", exist_ok=True)
    np.savez(tmp_path / "data" / "traj_a.npz", **{k[len("npz_"):]: d[k] for k in d.files if k.startswith("npz_")})
    return str(tmp_path / "data")


def _model(cfg, device):
    from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale
    return EncodeProcessDecodeMultiScale(node_in_dim=cfg.node_in_dim, edge_in_dim=cfg.edge_in_dim, hidden_size=cfg.hidden_size,
                                         process_steps=cfg.process_steps, coarse_process_steps=0, node_out_dim=cfg.node_out_dim,
                                         attention=False, sample_ratio=0.0, with_mat_params=cfg.with_mat_params,
                                         time_dim=cfg.time_dim, device=device).to(device)


def test_training_loop_of_the_reference_driver(tmp_path):
    from gnn_amd.datasetclass import HydrogelDataset
    from gnn_amd.loader import DataLoader                  # in place of torch_geometric.data.DataLoader (:5)
    data_dir = _write_trajectory(tmp_path)
    time_dim = 2
    cfg = O.OracleConfig(node_in_dim=3 + (time_dim + 1) + 5, process_steps=3, time_dim=time_dim)
    device = torch.device("cuda")
    dataset = HydrogelDataset(data_dir=data_dir, noise_level=0.01, add_targets=True, split_to_frames=True, time_dim=time_dim)
    torch.manual_seed(0)
    model = _model(cfg, device)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    norms = O.make_norms(cfg)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
    ref_opt = torch.optim.Adam(list(sd.values()), lr=1e-4, weight_decay=1e-5)
    model.train()
    steps = 0
    torch.manual_seed(77)                                   # noise draws of the dataset, then the loader's shuffle
    for traj_idx, data in enumerate(dataset):
        train_loader = DataLoader(data, batch_size=1, shuffle=True)
        seen = []
        for batch in train_loader:
            seen.append(float(batch["time"]))
            ref_opt.zero_grad()
            ref = O.loss(sd, cfg, norms, {k: batch[k] for k in batch.keys()})      # the same frame, on the host
            ref[0].backward()
            ref_opt.step()
            batch = batch.to(device)
            optimizer.zero_grad()
            loss, loss_ux, loss_uy, loss_phi = model.loss(batch)
            loss.backward()
            optimizer.step()
            got = [loss.item(), loss_ux.item(), loss_uy.item(), loss_phi.item()]
            want = [float(v.detach()) for v in ref]
            for a, b in zip(got, want):
                assert abs(a - b) <= 2e-4 * abs(b) + 1e-7, (steps, got, want)
            steps += 1
        assert sorted(seen) == [0.0, 0.5, 1.0, 1.5, 2.0] and seen != sorted(seen)     # every frame once, shuffled
    assert traj_idx == 0 and steps == 5
    # five Adam steps later the weights still agree.  Compared as one vector: Adam moves a coordinate whose
    # gradient is rounding noise by +-lr whatever its sign, so tensors that start at zero (LayerNorm biases)
    # cannot be compared one by one.
    mine = torch.cat([p.detach().cpu().reshape(-1) for _, p in model.named_parameters()])
    theirs = torch.cat([sd[k].detach().reshape(-1) for k, _ in model.named_parameters()])
    assert rel_l2(mine, theirs) < 1e-4, rel_l2(mine, theirs)
    assert float((mine - theirs).abs().max()) <= 2 * 5 * 1e-4 + 1e-6


def test_rollout_driver_sequence_from_a_checkpoint_directory(tmp_path):
    from gnn_amd.datasetclass import HydrogelDataset
    from gnn_amd.rollout import rollout
    data_dir = _write_trajectory(tmp_path)
    time_dim = 1
    cfg = O.OracleConfig(process_steps=4, time_dim=time_dim)
    device = torch.device("cuda")
    # a directory in the layout of selected_model/<name>/best_model, written by a first model
    torch.manual_seed(3)
    trained = _model(cfg, device)
    trained.output_normalizer(torch.randn(time_dim, 45, 3, device=device) * 1e-3)      # non-trivial statistics
    model_path = tmp_path / "selected_model" / "run" / "best_model"
    os.makedirs(model_path)
    trained.save_model(str(model_path))
    assert sorted(os.listdir(model_path)) == ["edge_features_normalizer.pth", "model_weights.pth",
                                              "node_features_normalizer.pth", "output_normalizer.pth"]
    model = _model(cfg, device)
    model.load_model(str(model_path))
    model.eval()
    dataset = HydrogelDataset(data_dir=data_dir, noise_level=0, time_dim=time_dim)
    for idx, data in enumerate(dataset):
        sample_name = dataset.get_name(idx)
        trajectory_rollout = rollout(model, data)
        saved = {k: trajectory_rollout[k].detach().cpu().numpy()
                 for k in ("time", "pred", "gt", "swelling_phi", "node_type", "cells", "mesh_pos", "rmse_x", "rmse_y", "rmse_phi")}
    assert idx == 0 and sample_name == "traj_a"
    T, n = len(data), data[0]["mesh_pos"].shape[0]
    assert saved["pred"].shape == (T, n, 3) and saved["gt"].shape == (T, n, 3) and saved["swelling_phi"].shape == (T, n)
    # the oracle walks core/rollout.py:5-45 with the checkpoint's weights and statistics
    sd = {k: v.detach().cpu() for k, v in trained.state_dict().items()}
    norms = O.make_norms(cfg)
    for key, nrm in (("node", trained.node_features_normalizer), ("edge", trained.edge_features_normalizer),
                     ("output", trained.output_normalizer)):
        norms[key] = O.NormStats(nrm._acc_sum.cpu().clone(), nrm._acc_sum_squared.cpu().clone(), nrm._acc_count, nrm._num_accumulations)
    static = {k: data[0][k] for k in ("mesh_pos", "node_type", "mat_param", "edge_index")}
    world = torch.stack([fr["world_pos"] for fr in data]); phi = torch.stack([fr["phi"] for fr in data])
    swell = torch.stack([fr["swelling_phi"] for fr in data])
    pred, rmse = O.rollout(sd, cfg, norms, static, world, phi, swell)
    assert rel_l2(torch.from_numpy(saved["pred"]), pred) < 1e-5
    got_rmse = np.array([saved["rmse_x"], saved["rmse_y"], saved["rmse_phi"]])
    assert np.allclose(got_rmse, rmse.numpy(), rtol=1e-4)
    # the normalisers kept accumulating during the rollout, as in the reference (core/normalization.py:19-24)
    assert model.node_features_normalizer._num_accumulations == norms["node"].num_accumulations
