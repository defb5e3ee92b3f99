"""Frame loader for the training loop.

The reference iterates a trajectory's frames with ``torch_geometric.data.DataLoader(data, batch_size=1,
shuffle=True)`` (run_train_multiscale.py:5, :135).  PyG is not a dependency here and its collater does not
know ``MeshGraph``, so this module provides the loader with the same constructor call: a
``torch.utils.data.DataLoader`` (hence the same sampler and the same shuffled order for the same seed as PyG's
loader, which is a subclass of it too) whose collate function hands a single frame through unchanged and
merges several frames into one block-diagonal frame.
"""
from __future__ import annotations

import torch.utils.data

from .graph import MeshGraph, batch_graphs


def collate_frames(frames):
    """batch_size 1: the frame itself (its fields are what ``model.loss`` reads); more: the block-diagonal
    union of the frames (``gnn_amd.graph.batch_graphs``; they must agree on ``mat_param``)."""
    frames = list(frames)
    if not all(isinstance(f, MeshGraph) for f in frames):
        raise TypeError("gnn_amd.loader.DataLoader batches MeshGraph frames")
    return frames[0] if len(frames) == 1 else batch_graphs(frames)


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        kwargs.pop("follow_batch", None)          # PyG-only options, meaningless for MeshGraph frames
        kwargs.pop("exclude_keys", None)
        kwargs.setdefault("collate_fn", collate_frames)
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, **kwargs)
