"""GPU: the coarse graph of the multiscale option built on the device (core/meshgraphnet.py:237-283) -- ``mgn_fps`` and
``mgn_knn_graph`` through the C ABI -- against the host definitions of gnn_amd/coarse.py, index for index.

Sampling: the host function is the reference's own torch expressions on the CPU (:243-254), pinned by fixture G7's
sampled ids; regular grids make most distances tie, so the float32 rounding of the distance and the lowest-index rule of
``torch.argmax`` both decide the result.  Nearest neighbours: ``knn_graph`` is torch-cluster's in the reference, absent
from its tree and this image; device kernel, host function and oracle share one restatement (parity unpinned)."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import GOLDEN
from gnn_amd.coarse import farthest_point_sampling, knn_graph, sample_and_connect

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grid(nx, ny, dx=0.08 / 32, dy=0.01 / 16):
    xs, ys = torch.meshgrid(torch.arange(nx, dtype=torch.float32) * dx, torch.arange(ny, dtype=torch.float32) * dy, indexing="xy")
    return torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).contiguous()


def _cloud(n, dim, seed):
    return torch.rand(n, dim, generator=torch.Generator().manual_seed(seed))


SAMPLING = [
    ("bending grid, every distance tied many times", lambda: _grid(33, 17), 280, 17),
    ("unit grid", lambda: _grid(40, 25, 1.0, 1.0), 500, 999),
    ("one point per thread", lambda: _cloud(1000, 2, 1), 400, 3),
    ("four points per thread, 3-D", lambda: _cloud(3000, 3, 2), 300, 2999),
    ("sixteen points per thread", lambda: _cloud(10000, 2, 3), 200, 0),
    ("distances in device memory", lambda: _cloud(20000, 2, 4), 150, 12345),
    ("distances in device memory, 3-D grid-like", lambda: torch.cat([_grid(150, 120, 1.0, 1.0), torch.zeros(18000, 1)], 1), 120, 7),
    ("more samples than distinct points", lambda: torch.cat([_cloud(30, 2, 5)] * 2), 45, 1),
    ("two launches per sample over the whole GPU", lambda: _cloud(40000, 2, 6), 150, 39999),
    ("two launches per sample, regular grid", lambda: _grid(250, 160, 1.0, 1.0), 130, 0),
    ("two launches per sample, 3-D", lambda: _cloud(70001, 3, 7), 90, 5),
]


@pytest.mark.parametrize("what,make,m,start", SAMPLING, ids=[c[0] for c in SAMPLING])
def test_device_sampling_equals_the_reference_expressions(what, make, m, start):
    from gnn_amd import ops
    pts = make()
    n = pts.shape[0]
    got = ops.fps(pts.to(DEV), [n], [m], [start]).cpu()
    want = farthest_point_sampling(pts, m, start)
    assert got.dtype == torch.int64 and torch.equal(got, want), (what, int((got != want).nonzero()[0]))


def test_both_sampling_kernels_agree_on_a_large_mesh():
    """The single-workgroup kernel (what a batch of meshes uses) and the per-sample launch sequence on the same
    100 000 points: the same samples."""
    import ctypes as C
    from gnn_amd import _lib, ops
    pts = _cloud(100000, 2, 9).to(DEV)
    n, m = pts.shape[0], 300
    by_rounds = ops.fps(pts, [n], [m], [77])
    ws = torch.empty(n, dtype=torch.float32, device=DEV)
    one_block = torch.empty(m, dtype=torch.int64, device=DEV)
    ptrs = [torch.tensor(v, dtype=torch.int32, device=DEV) for v in ([0, n], [0, m], [77])]
    rc = _lib.lib().mgn_fps(pts.data_ptr(), 2, ptrs[0].data_ptr(), ptrs[1].data_ptr(), ptrs[2].data_ptr(), 1, n, ws.data_ptr(),
                            one_block.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "mgn_fps")
    assert torch.equal(by_rounds, one_block)


@pytest.mark.parametrize("tag,ratio", [("ms", 0.5), ("both", 0.4)])
def test_device_sampling_and_graph_golden(tag, ratio):
    """The sampled ids of the reference itself (fixture G7) and the coarse edges stored with them."""
    d = np.load(f"{GOLDEN}/g7_attention_multiscale.npz")
    g = O.synthetic_graph(int(d["nx"]), int(d["ny"]), seed=int(d["graph_seed"]))
    first, ei, counts = sample_and_connect(g["mesh_pos"].to(DEV), ratio, [int(d[f"{tag}_start1"])])
    assert first.is_cuda and ei.is_cuda and counts == [int(g["mesh_pos"].shape[0] * ratio)]
    assert np.array_equal(first.cpu().numpy(), d[f"{tag}_first1"])
    assert np.array_equal(ei.cpu().numpy(), d[f"{tag}_coarse_edge_index1"])


def test_batched_sampling_and_graph_equal_block_by_block():
    """Blocks of unequal size, one workgroup each; block b's ids and edges are its stand-alone result shifted."""
    from gnn_amd import ops
    blocks = [_grid(33, 17), _cloud(1500, 2, 7), _cloud(40, 2, 8), _grid(5, 1, 1.0, 1.0)]
    sizes = [b.shape[0] for b in blocks]
    starts = [5, 1499, 0, 4]
    pts = torch.cat(blocks).to(DEV)
    first, ei, counts = sample_and_connect(pts, 0.37, starts, sizes)
    h_first, h_ei, h_counts = sample_and_connect(torch.cat(blocks), 0.37, starts, sizes)
    assert counts == h_counts == [max(1, int(s * 0.37)) for s in sizes]
    assert torch.equal(first.cpu(), h_first) and torch.equal(ei.cpu(), h_ei)
    n0 = c0 = 0
    for b, s, m, st in zip(blocks, sizes, counts, starts):
        alone = farthest_point_sampling(b, m, st)
        assert torch.equal(first[c0: c0 + m].cpu(), alone + n0)
        sel = (h_ei[1] >= c0) & (h_ei[1] < c0 + m)
        assert torch.equal(h_ei[:, sel], knn_graph(b[alone], 3) + c0)
        n0, c0 = n0 + s, c0 + m
    with pytest.raises(RuntimeError):
        ops.fps(pts, sizes, counts, [5, 1500, 0, 4])            # a first sample outside its block


@pytest.mark.parametrize("n,dim,k,seed", [(1, 2, 3, 0), (2, 2, 3, 0), (4, 2, 3, 1), (300, 2, 3, 2), (257, 3, 4, 3), (5000, 2, 3, 4),
                                          (700, 3, 8, 5)])
def test_device_knn_graph_equals_host_on_point_clouds(n, dim, k, seed):
    from gnn_amd import ops
    pts = _cloud(n, dim, seed)
    got = ops.knn_graph(pts.to(DEV), None, [n], k).cpu()
    assert torch.equal(got, knn_graph(pts, k)) and torch.equal(got, O.knn_graph(pts, k))


def test_device_knn_graph_ties_and_indirection():
    """Ties by lower index on a unit grid and on coincident points; coarse points addressed through ``first``."""
    from gnn_amd import ops
    pts = _grid(9, 7, 1.0, 1.0)
    ei = ops.knn_graph(pts.to(DEV), None, [63], 3).cpu()
    assert torch.equal(ei, knn_graph(pts, 3))
    zeros = torch.zeros(40, 2)
    ei0 = ops.knn_graph(zeros.to(DEV), None, [40], 3).cpu()
    assert torch.equal(ei0, knn_graph(zeros, 3))
    assert ei0[0, ei0[1] == 20].tolist() == [0, 1, 2] and ei0[0, ei0[1] == 0].tolist() == [1, 2, 3]
    cloud = _cloud(900, 2, 11)
    first = torch.randperm(900, generator=torch.Generator().manual_seed(1))[:333]
    got = ops.knn_graph(cloud.to(DEV), first.to(DEV), [333], 3).cpu()
    assert torch.equal(got, knn_graph(cloud[first], 3))
    with pytest.raises(RuntimeError):
        ops.knn_graph(cloud.to(DEV), None, [900], 5)            # k outside the instantiated set


def test_model_builds_its_coarse_graph_on_the_device():
    from test_gpu_options import build
    cfg = O.OracleConfig(process_steps=1, sample_ratio=0.5, coarse_process_steps=1)
    model = build(cfg, O.init_state_dict(cfg, 3))
    g = O.synthetic_graph(33, 17, seed=1)
    model.coarse_start = 100
    cg = model._coarse_graph({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in g.items()})
    assert cg.first.is_cuda and cg.edge_index.is_cuda and cg.topology.rowptr.is_cuda
    first = farthest_point_sampling(g["mesh_pos"].float(), int(g["mesh_pos"].shape[0] * 0.5), 100)
    assert torch.equal(cg.first_cpu, first)
    assert torch.equal(cg.edge_index.cpu(), knn_graph(g["mesh_pos"].float()[first], 3))
