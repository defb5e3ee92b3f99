"""gnn_amd: the MeshGraphNets encode-process-decode hot path of narupanta/gnn on AMD
MI355X (gfx950) -- hand-written HIP kernels behind the reference's Python API.

    from gnn_amd.meshgraphnet import EncodeProcessDecodeMultiScale
    from gnn_amd.rollout import rollout
    from gnn_amd.datasetclass import cells_to_edge_index

The arithmetic lives in ``libmgn_hip.so`` (C ABI: ``include/mgn_hip.h``); importing this
package does not load it, the first kernel call does and raises if it is missing.
"""
from .graph import MeshGraph, Topology, batch_graphs, unbatch_nodes  # noqa: F401
from .normalization import Normalizer  # noqa: F401
from .meshgraphnet import MLP, EdgeNodeMessagePassing, EncodeProcessDecodeMultiScale  # noqa: F401
from .datasetclass import cells_to_edge_index, grid_mesh, strip_local_graph, synthetic_frames, synthetic_strip  # noqa: F401
from .rollout import rollout, rollout_batched  # noqa: F401

__version__ = "0.1.0"
