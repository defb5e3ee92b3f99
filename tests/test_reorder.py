"""CPU: locality reordering (gnn_amd/reorder.py).  A mesh whose node ids were shuffled gets a
numbering with small neighbour distance back; the model is equivariant under the renumbering,
so the oracle's prediction on the permuted frame, restored, equals its prediction on the
original frame (tolerance 1e-5: the receiver sums add in a different order)."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import rel_l2


def _shuffled(nx, ny, seed):
    from gnn_amd import reorder as R
    g = O.synthetic_graph(nx, ny, seed=seed)
    n = g["mesh_pos"].shape[0]
    shuffle = np.random.default_rng(seed).permutation(n)
    return R.permute_frame(g, shuffle), n


@pytest.mark.parametrize("method", ["rcm", "morton"])
def test_order_restores_locality(method):
    from gnn_amd import reorder as R
    from gnn_amd.partition import HaloPlan
    g, n = _shuffled(40, 25, 1)
    ei = g["edge_index"].numpy()
    order = R.node_order(g, method)
    assert np.array_equal(np.sort(order), np.arange(n))
    gp = R.permute_frame(g, order)
    eip = gp["edge_index"].numpy()
    # same mesh: relabelling back gives the same edge list, and node fields travel with their node
    assert np.array_equal(order[eip], ei)
    assert torch.equal(gp["mesh_pos"], g["mesh_pos"][torch.from_numpy(order)])
    assert torch.equal(R.restore_nodes(gp["phi"], order), g["phi"])
    assert np.array_equal(order[gp["cells"].numpy()], g["cells"].numpy())
    assert R.bandwidth(ei) > n // 2
    if method == "rcm":
        assert R.bandwidth(eip) <= 2 * 25 + 2, R.bandwidth(eip)          # about one mesh row
    mean_dist = lambda e: float(np.abs(e[0] - e[1]).mean())
    assert mean_dist(eip) * 8 < mean_dist(ei)          # the Z-order curve jumps at quadrant borders, but rarely
    # what the contiguous-range partitioner gains: halo of rank 1 of 4
    before = HaloPlan(ei, n, 4, 1).num_halo
    after = HaloPlan(eip, n, 4, 1).num_halo
    assert after * 3 < before, (before, after)


def test_prediction_is_equivariant_under_renumbering():
    from gnn_amd import reorder as R
    cfg = O.OracleConfig(process_steps=3)
    sd = O.init_state_dict(cfg, 2)
    g, n = _shuffled(14, 9, 3)
    order = R.node_order(g, "rcm")
    with torch.no_grad():
        ref = O.predict(sd, cfg, O.make_norms(cfg), g)
        out = O.predict(sd, cfg, O.make_norms(cfg), R.permute_frame(g, order))
    assert rel_l2(R.restore_nodes(out, order, dim=1), ref) < 1e-5


def test_permutation_size_is_checked():
    from gnn_amd import reorder as R
    g = O.synthetic_graph(5, 4)
    with pytest.raises(ValueError):
        R.permute_frame(g, np.arange(7))
    with pytest.raises(ValueError):
        R.node_order(g, "hilbert")
