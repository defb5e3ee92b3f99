"""CPU: the C-ABI library loads, exports every symbol the header declares, and its host
routines (edge build, receiver-major CSR) agree with the oracle exactly.  No kernel is
launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import oracle as O
from conftest import GOLDEN, ROOT


def _lib_or_build():
    from gnn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from gnn_amd.csrc.build import build
        build()
    return _lib


def test_header_symbols_exported():
    _lib = _lib_or_build()
    hdr = open(os.path.join(ROOT, "include", "mgn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mgn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 17
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mgn_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib().mgn_abi_version() == 1


def test_cells_to_edge_index_matches_golden_and_oracle():
    from gnn_amd.datasetclass import cells_to_edge_index
    d = np.load(f"{GOLDEN}/g1_edges.npz")
    for name in ("g2x2", "g4x3", "g32x16", "irregular", "two_tris"):
        ei = cells_to_edge_index(torch.from_numpy(d[f"{name}_cells"]))
        assert ei.dtype == torch.int64 and np.array_equal(ei.numpy(), d[f"{name}_edge_index"]), name
    rng = np.random.default_rng(0)
    for _ in range(5):                                   # random soups of triangles, ids up to 1000
        cells = rng.integers(0, 1000, size=(rng.integers(1, 400), 3))
        assert np.array_equal(cells_to_edge_index(torch.from_numpy(cells)).numpy(), O.cells_to_edge_index(cells))
    assert cells_to_edge_index(torch.zeros(0, 3, dtype=torch.long)).shape == (2, 0)


def test_cells_to_edge_index_large_grid():
    from gnn_amd.datasetclass import cells_to_edge_index, grid_mesh
    pos, cells = grid_mesh(316, 316)                      # the 100k-node synthetic mesh
    ei = cells_to_edge_index(torch.from_numpy(cells)).numpy()
    assert pos.shape[0] == 100489 and ei.shape == (2, 600400)
    assert np.array_equal(ei, O.cells_to_edge_index(cells))
    p2, c2 = O.grid_mesh(316, 316)
    assert np.array_equal(cells, c2) and np.allclose(pos, p2, atol=1e-7)


def test_cells_to_edge_index_rejects_negative_ids():
    from gnn_amd.datasetclass import cells_to_edge_index
    with pytest.raises(RuntimeError, match="negative"):
        cells_to_edge_index(torch.tensor([[0, 1, -2]]))


def test_receiver_csr_matches_oracle():
    from gnn_amd import ops
    rng = np.random.default_rng(1)
    for n, e in ((1, 0), (5, 3), (50, 400), (1000, 9000)):
        ei = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)])
        got = ops.receiver_csr_np(ei, n)
        exp = O.receiver_csr(ei, n)
        for a, b in zip(got, exp):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    with pytest.raises(RuntimeError, match="outside"):
        ops.receiver_csr_np(np.array([[0, 5], [1, 1]]), 3)


def test_no_cpu_fallback():
    from gnn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.mp_project(torch.zeros(4, 128), torch.zeros(128, 384), torch.zeros(128), torch.zeros(4, 128), torch.zeros(4, 128))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_state_dict_layout_and_seeded_init():
    from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale
    cfg = O.OracleConfig()
    torch.manual_seed(0)
    m = EncodeProcessDecodeMultiScale(10, 7, hidden_size=128, process_steps=10, coarse_process_steps=0, node_out_dim=3,
                                      attention=False, with_mat_params=False, sample_ratio=0.0, time_dim=1, device="cpu")
    sd, ref = m.state_dict(), O.init_state_dict(cfg, 0)
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 136
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k
    assert sum(p.numel() for p in m.parameters()) == 1209987


def test_normalizer_pickle_layout(tmp_path):
    """Saved normalisers carry the reference's class path; the tensor-only unpickler reads them back."""
    import pickletools, zipfile
    from gnn_amd.normalization import Normalizer, load_normalizer
    n = Normalizer(1, 7, "edge_features_normalizer", "cpu")
    n._acc_sum += 3.0; n._acc_count = 12; n._num_accumulations = 2
    torch.save(n, tmp_path / "n.pth")
    with zipfile.ZipFile(tmp_path / "n.pth") as z:
        pkl = [x for x in z.namelist() if x.endswith("data.pkl")][0]
        ops_ = [(op.name, arg) for op, arg, _ in pickletools.genops(z.read(pkl))]
    assert ("GLOBAL", "core.normalization Normalizer") in ops_
    back = load_normalizer(tmp_path / "n.pth", "cpu")
    assert back._acc_count == 12 and back._num_accumulations == 2 and torch.equal(back._acc_sum, n._acc_sum)
    assert back._name == "edge_features_normalizer" and back._max_accumulations == 10 ** 6


def test_mesh_graph_container():
    import copy
    from gnn_amd.graph import MeshGraph
    g = MeshGraph(a=torch.zeros(3), k=5)
    assert g["a"] is g.a and "a" in g and g.k == 5
    h = copy.deepcopy(g); h.a += 1
    assert float(g.a.sum()) == 0.0
    g.b = torch.ones(2); assert g["b"].shape == (2,)
    assert g.to("cpu")["a"].device.type == "cpu"


def test_batch_graphs_block_diagonal():
    """Batched frames: node fields concatenated, node ids shifted per block, nothing shared."""
    import gnn_amd
    a = gnn_amd.synthetic_frames(3, 2, seed=1)[0]
    b = gnn_amd.synthetic_frames(4, 3, seed=2)[0]
    g = gnn_amd.batch_graphs([a, b])
    na, nb = a["mesh_pos"].shape[0], b["mesh_pos"].shape[0]
    assert g["batch_sizes"] == [na, nb] and g["mesh_pos"].shape[0] == na + nb
    ea = a["edge_index"].shape[1]
    assert torch.equal(g["edge_index"][:, :ea], a["edge_index"]) and torch.equal(g["edge_index"][:, ea:], b["edge_index"] + na)
    assert int(g["edge_index"][:, :ea].max()) < na <= int(g["edge_index"][:, ea:].min())
    assert torch.equal(g["cells"][a["cells"].shape[0]:], b["cells"] + na)
    parts = gnn_amd.unbatch_nodes(g["phi"], g["batch_sizes"])
    assert torch.equal(parts[0], a["phi"]) and torch.equal(parts[1], b["phi"])
    b2 = gnn_amd.synthetic_frames(4, 3, seed=2)[0]
    b2["mat_param"] = b2["mat_param"] * 2
    with pytest.raises(ValueError):
        gnn_amd.batch_graphs([a, b2])


def test_size_limits_are_rejected_on_the_host():
    """Maximum sizes: arguments beyond what the kernels address are refused before anything is launched (no GPU
    needed: the pointers are never followed).  Node rows are addressed with 32-bit element offsets (n * 128 < 2^31),
    edge and node counts are 31-bit, decoder outputs are at most 32 wide, feature rows at most 16."""
    import ctypes as C
    from gnn_amd import _lib
    lib = _lib.lib()
    p = C.c_void_p(0x1000)
    mlp = _lib.MgnMlp(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000)
    n_max = (1 << 31) // 128
    assert lib.mgn_mp_edge(p, p, p, p, p, p, C.byref(mlp), p, p, 10, n_max, None) == -1
    assert b"mgn_mp_edge" in lib.mgn_last_error()
    assert lib.mgn_mp_edge(p, p, p, p, p, p, C.byref(mlp), p, p, 1 << 31, 10, None) == -1
    assert lib.mgn_mp_node(p, p, p, C.byref(mlp), p, 1 << 31, None) == -1
    assert lib.mgn_mp_edge(p, p, p, p, p, p, C.byref(mlp), p, p, -1, 10, None) == -1
    assert lib.mgn_mp_edge(None, p, p, p, p, p, C.byref(mlp), p, p, 10, 10, None) == -1
    assert lib.mgn_feature_sums(p, 10, 17, p, p, None) == -1                      # wider than MGN_MAX_FEAT
    assert lib.mgn_encode(p, 17, p, C.byref(mlp), p, 10, None) == -1
    assert lib.mgn_decode(p, C.byref(mlp), 11, 3, p, None, None, None, None, None, 10, None) == -1   # 33 outputs
    # empty inputs are not an error and launch nothing
    assert lib.mgn_mp_edge(None, None, None, None, None, None, C.byref(mlp), None, None, 0, 0, None) == 0
    assert lib.mgn_mp_node(None, None, None, C.byref(mlp), None, 0, None) == 0
    # backward of the attention option: same conventions
    assert lib.mgn_train_attn_edge_bwd(p, p, p, p, p, None, p, p, 1 << 31, None) == -1
    assert b"mgn_train_attn_edge_bwd" in lib.mgn_last_error()
    assert lib.mgn_train_attn_edge_bwd(None, p, p, p, p, None, p, p, 10, None) == -1
    assert lib.mgn_train_attn_alpha_bwd(p, p, p, None, 10, None) == -1
    assert lib.mgn_train_attn_alpha_bwd(p, p, p, p, 1 << 31, None) == -1
    assert lib.mgn_train_attn_node_grad(p, p, None, None, p, p, 0, 10, None) == -1
    assert lib.mgn_train_attn_edge_bwd(None, None, None, None, None, None, None, None, 0, None) == 0
    assert lib.mgn_train_attn_alpha_bwd(None, None, None, None, 0, None) == 0
    assert lib.mgn_train_attn_node_grad(None, None, None, None, None, None, 0, 0, None) == 0


def test_argument_checks_of_the_coarse_graph_feedback_and_reduction_entry_points():
    """The entry points added for the coarse graph, the folded launches and the deferred reductions refuse bad
    arguments on the host (nothing is launched, no pointer is followed) and accept empty inputs."""
    import ctypes as C
    from gnn_amd import _lib
    lib = _lib.lib()
    p = C.c_void_p(0x1000)
    mlp = _lib.MgnMlp(0x1000, 0x1000, 0x1000, 0x1000, 0, 0)
    # farthest-point sampling / nearest neighbours
    assert lib.mgn_fps(p, 4, p, p, p, 1, 10, None, p, None) == -1 and b"mgn_fps" in lib.mgn_last_error()      # dim 4
    assert lib.mgn_fps(p, 2, p, p, p, 1, 20000, None, p, None) == -1          # beyond the register variants without a workspace
    assert lib.mgn_fps(p, 2, None, p, p, 1, 10, None, p, None) == -1
    assert lib.mgn_fps(None, 2, None, None, None, 0, 0, None, None, None) == 0
    assert lib.mgn_fps_workspace_floats(1000) >= 1000 and lib.mgn_fps_workspace_floats(-1) == -1
    assert lib.mgn_fps_single(p, 2, 100, 10, 100, p, p, None) == -1           # first sample outside the mesh
    assert lib.mgn_fps_single(p, 2, 100, 10, 5, None, p, None) == -1          # no workspace
    assert lib.mgn_fps_single(p, 2, 100, 0, 5, None, None, None) == 0         # nothing to sample
    assert lib.mgn_knn_graph(p, 2, None, p, p, 1, 10, 5, p, p, None) == -1 and b"k = 5" in lib.mgn_last_error()
    assert lib.mgn_knn_graph(p, 2, None, p, p, 1, 10, 0, p, p, None) == -1
    assert lib.mgn_knn_graph(p, 2, None, p, p, 70000, 10, 3, p, p, None) == -1                                  # more blocks than a grid dimension holds
    assert lib.mgn_knn_graph(None, 2, None, None, None, 0, 0, 3, None, None, None) == 0
    # features with sums, folded statistics update, decoder with state feedback
    assert lib.mgn_node_features_sums(p, p, p, p, 2, p, None, p, 10, None, None) == -1                          # no partial-sum space
    assert lib.mgn_node_features_sums(p, p, p, p, 9, p, None, p, 10, p, None) == -1                             # 17 feature columns
    assert lib.mgn_edge_features_sums(p, p, p, p, p, p, 1 << 31, p, None) == -1
    assert lib.mgn_norm_fold(p, 10, 17, p, p, None, 10, 1e-8, p, None) == -1
    assert lib.mgn_norm_fold(None, 10, 7, p, p, None, 10, 1e-8, p, None) == -1
    assert lib.mgn_decode_step(p, C.byref(mlp), 11, p, p, p, p, p, p, p, 10, None) == -1                        # 33 outputs
    assert lib.mgn_decode_step(p, C.byref(mlp), 1, p, p, p, p, p, None, p, 10, None) == -1                      # no next-state buffer
    assert lib.mgn_decode_step(None, C.byref(mlp), 1, None, None, None, None, None, None, None, 0, None) == 0
    # deferred reductions: a record list needs room and a counter; records are validated before the launch
    recs = (_lib.MgnReduceRec * 2)()
    cnt = C.c_int32(0)
    assert lib.mgn_train_defer_reductions(recs, 0, C.byref(cnt)) == -1
    assert lib.mgn_train_defer_reductions(recs, 2, None) == -1
    cnt.value = 3
    assert lib.mgn_train_defer_reductions(recs, 2, C.byref(cnt)) == -1                                          # count beyond the capacity
    cnt.value = 0
    assert lib.mgn_train_defer_reductions(recs, 2, C.byref(cnt)) == 0 and lib.mgn_train_defer_reductions(None, 0, None) == 0
    assert lib.mgn_train_reduce_many(recs, 1, None) == -1                                                       # an empty record
    assert lib.mgn_train_reduce_many(None, 0, None) == 0
    assert lib.mgn_train_partial_bytes(3, 10) == -1 and lib.mgn_train_partial_bytes(0, -1) == -1
