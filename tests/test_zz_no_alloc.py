"""mgn_feature_sums_static: the entry point that allocates nothing (partial sums in a buffer inside
the library) against mgn_feature_sums with a caller-provided workspace.  Integer-exact comparison:
both run the same kernels with the same block partition, so the float64 sums must be identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(rows, n_feat, seed):
    from gnn_amd import ops
    g = torch.Generator().manual_seed(seed)
    feat = (torch.randn(rows, n_feat, generator=g) * 3.0 + 0.5).cuda()
    ref = torch.full((2 * n_feat,), float("nan"), dtype=torch.float64, device="cuda")
    scratch = torch.empty(ops.feature_sums_scratch_bytes(rows, n_feat), dtype=torch.uint8, device="cuda")
    ops.feature_sums(feat, ref, scratch)
    out = torch.full((2 * n_feat,), float("nan"), dtype=torch.float64, device="cuda")
    ops.feature_sums_static(feat, out)
    torch.cuda.synchronize()
    return feat, ref, out


def test_static_sums_equal_workspace_sums_two_calls_in_a_row():
    # first call: large and wide (fills many partial rows and all 16 columns of the library's buffer)
    feat, ref, out = _both(300_007, 16, 1)
    assert torch.equal(out, ref)
    f64 = feat.double()
    exact = torch.cat([f64.sum(0), (f64 * f64).sum(0)])
    assert float((out - exact).abs().max() / exact.abs().max()) < 1e-12
    # second call right after it: fewer rows and fewer columns -- nothing of the first call may leak in
    feat, ref, out = _both(777, 7, 2)
    assert torch.equal(out, ref)
    f64 = feat.double()
    exact = torch.cat([f64.sum(0), (f64 * f64).sum(0)])
    assert float((out - exact).abs().max() / exact.abs().max()) < 1e-12
    # and a single row
    feat, ref, out = _both(1, 3, 3)
    assert torch.equal(out, ref)


def test_static_sums_reject_oversize_and_bad_arguments():
    from gnn_amd import _lib
    lib = _lib.lib()
    sums = torch.zeros(32, dtype=torch.float64, device="cuda")
    feat = torch.zeros(4, 4, device="cuda")
    too_many = (1 << 31) + 1
    rc = lib.mgn_feature_sums_static(feat.data_ptr(), too_many, 4, sums.data_ptr(), None)
    assert rc == -3 and b"exceed the limit" in lib.mgn_last_error()
    assert lib.mgn_feature_sums_static(feat.data_ptr(), 4, 17, sums.data_ptr(), None) == -1
    assert lib.mgn_feature_sums_static(feat.data_ptr(), 4, 4, None, None) == -1
    torch.cuda.synchronize()
    assert float(sums.abs().sum()) == 0.0


def test_static_sums_inside_a_captured_graph():
    """The use the entry point exists for: called under HIP graph capture (after the one-time address look-up) and
    replayed on new data, it gives the sums of the workspace version."""
    from gnn_amd import ops
    ops.feature_sums_static_prepare()
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(50_000, 10, generator=g).cuda()
    out = torch.zeros(20, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.feature_sums_static(feat, out)             # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.feature_sums_static(feat, out)
    for seed in (10, 11):
        feat.copy_(torch.randn(50_000, 10, generator=torch.Generator().manual_seed(seed)))
        graph.replay()
        ref = torch.empty(20, dtype=torch.float64, device="cuda")
        scratch = torch.empty(ops.feature_sums_scratch_bytes(50_000, 10), dtype=torch.uint8, device="cuda")
        ops.feature_sums(feat, ref, scratch)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
