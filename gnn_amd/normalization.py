"""Online feature normaliser with device-resident statistics.

Mirrors ``core/normalization.py:4-51`` of the reference: same constructor, same attribute
names (``_acc_sum``, ``_acc_sum_squared``, ``_acc_count``, ``_num_accumulations``,
``_max_accumulations``, ``_std_epsilon``, ``_name``, ``_device``), same behaviour --
``forward`` accumulates before it normalises and keeps accumulating in eval mode until
``_max_accumulations`` updates have been seen.  The sums live on the GPU and are updated
by ``mgn_feature_sums`` + ``mgn_norm_update``; the counts stay Python ints.

The class pickles under the reference's module path (``core.normalization.Normalizer``)
so ``save_model`` / ``load_model`` keep the reference's checkpoint layout.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn

from . import ops


class Normalizer(nn.Module):
    def __init__(self, batch_size, feature_size, name, device, max_accumulations=10 ** 6, std_epsilon=1e-8):
        super().__init__()
        self._name = name
        self._device = device
        self._max_accumulations = max_accumulations
        self._std_epsilon = std_epsilon
        self._acc_count = 0
        self._num_accumulations = 0
        self._acc_sum = torch.zeros((batch_size, feature_size), dtype=torch.float, device=device)
        self._acc_sum_squared = torch.zeros((batch_size, feature_size), dtype=torch.float, device=device)

    # ----------------------------------------------------------------- internals
    def _work(self):
        """Per-instance device scratch, created lazily (never pickled)."""
        w = self.__dict__.get("_mgn_work")
        dev = self._acc_sum.device
        if w is None or w["dev"] != dev:
            f = self._acc_sum.numel()
            w = {"dev": dev, "mean_std": torch.empty(2 * f, dtype=torch.float32, device=dev),
                 "sums": torch.empty(2 * f, dtype=torch.float64, device=dev), "scratch": None}
            self.__dict__["_mgn_work"] = w
        return w

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_mgn_work", None)
        return d

    def _require_device(self):
        if not self._acc_sum.is_cuda:
            raise RuntimeError(f"Normalizer '{self._name}': statistics are on {self._acc_sum.device}; the HIP path "
                               "needs them on a GPU (construct the model with device='cuda' or call to_device)")

    def to_device(self, device):
        """Moves the statistics (plain attributes, so ``nn.Module.to`` does not see them)."""
        self._acc_sum = self._acc_sum.to(device)
        self._acc_sum_squared = self._acc_sum_squared.to(device)
        self._device = device
        return self

    def use_device_count(self, on: bool = True):
        """Keeps a copy of ``_acc_count`` in device memory and lets the update kernel advance it, so
        that a captured (HIP graph) launch sequence stays valid from step to step.  ``_acc_count``
        itself is still advanced on the host by every call that goes through :meth:`stats_for`; a
        caller that replays captured launches advances it with :meth:`count_replayed`."""
        w = self._work()
        if on:
            w["count_dev"] = torch.tensor([self._acc_count], dtype=torch.int64, device=self._acc_sum.device)
        else:
            w.pop("count_dev", None)
        return self

    def count_replayed(self, rows: int):
        """Host bookkeeping for one replay of a captured accumulating call (core/normalization.py:36-37)."""
        self._acc_count += int(rows)
        self._num_accumulations += 1

    def _rows_view(self, batched_data: torch.Tensor) -> torch.Tensor:
        """[B, n, F] -> [n, B*F] rows whose column sums are the reference's sum over dim 1."""
        b, n, f = batched_data.shape
        if (b, f) != tuple(self._acc_sum.shape):
            raise RuntimeError(f"Normalizer '{self._name}': data [B={b}, n, F={f}] does not match statistics "
                               f"{tuple(self._acc_sum.shape)}")
        if b == 1:
            return batched_data.reshape(n, f)
        return batched_data.permute(1, 0, 2).reshape(n, b * f)

    def local_sums(self, rows2d: torch.Tensor, accumulate: bool = True, out: torch.Tensor = None):
        """First half of :meth:`stats_for`: when accumulation is live, the float64 column sums
        ``[sum | sum of squares]`` of ``rows2d`` [n, B*F] (into ``out`` [2*B*F] if given, else into the
        instance's own buffer); returns that tensor, or None when nothing is to be accumulated."""
        self._require_device()
        if not (accumulate and self._num_accumulations < self._max_accumulations):
            return None
        w = self._work()
        n, bf = rows2d.shape
        need = ops.feature_sums_scratch_bytes(n, bf)
        if w["scratch"] is None or w["scratch"].numel() * 8 < need:
            w["scratch"] = torch.empty((need + 7) // 8, dtype=torch.float64, device=rows2d.device)
        sums = w["sums"] if out is None else out
        ops.feature_sums(rows2d, sums, w["scratch"])
        return sums

    def accumulating(self) -> bool:
        return self._num_accumulations < self._max_accumulations

    def partials_for(self, rows: int) -> torch.Tensor:
        """Float64 buffer for the per-block partial sums of ``rows`` feature rows (``ops.node_features`` /
        ``ops.edge_features`` with ``partials=``), to be folded by :meth:`fold_partials`."""
        self._require_device()
        w = self._work()
        need = ops.feature_sums_scratch_bytes(rows, self._acc_sum.numel())
        if w["scratch"] is None or w["scratch"].numel() * 8 < need:
            w["scratch"] = torch.empty((need + 7) // 8, dtype=torch.float64, device=self._acc_sum.device)
        return w["scratch"]

    def fold_partials(self, rows: int) -> torch.Tensor:
        """:meth:`stats_for` for rows whose partial sums are already in :meth:`partials_for`'s buffer: one launch adds
        them, updates the statistics (core/normalization.py:30-37) and writes ``[mean | std]`` (:39-48)."""
        w = self._work()
        self._acc_count += int(rows)
        self._num_accumulations += 1
        w["fresh"] = None
        ops.norm_fold(w["scratch"], int(rows), self._acc_sum, self._acc_sum_squared, w.get("count_dev"), self._acc_count,
                      self._std_epsilon, w["mean_std"])
        return w["mean_std"]

    def _stats_key(self):
        a, b = self._acc_sum, self._acc_sum_squared
        return (id(a), id(b), a._version, b._version, a.data_ptr(), b.data_ptr(), self._acc_count, self._std_epsilon)

    def fold(self, sums, rows: int) -> torch.Tensor:
        """Second half of :meth:`stats_for`: folds ``sums`` (from :meth:`local_sums`, possibly reduced over
        ranks since; None = nothing to accumulate) over ``rows`` rows into the statistics
        (core/normalization.py:30-37) and returns the device vector ``[mean | std]`` (:39-48)."""
        self._require_device()
        w = self._work()
        added = 0
        if sums is not None:
            added = int(rows)
            self._acc_count += added
            self._num_accumulations += 1
            w["fresh"] = None
        else:
            # nothing accumulates: [mean | std] of unchanged statistics is still in the buffer.  The statistics change
            # only through the kernels launched here (each of which rewrites the buffer), through replaced tensors or
            # through in-place torch operations; with the row count in device memory, host and device counts move together
            key = self._stats_key()
            hit = w.get("fresh")
            if hit is not None and hit[0] == key:
                return w["mean_std"]
            w["fresh"] = (key, self._acc_sum, self._acc_sum_squared)     # the tensors stay alive: their ids cannot be reused
        if w.get("count_dev") is not None:
            ops.norm_update_counted(self._acc_sum, self._acc_sum_squared, sums, w["count_dev"], added, self._std_epsilon,
                                    w["mean_std"])
        else:
            ops.norm_update(self._acc_sum, self._acc_sum_squared, sums, self._acc_count, self._std_epsilon, w["mean_std"])
        return w["mean_std"]

    def stats_for(self, rows2d: torch.Tensor, accumulate: bool = True, partial_sums_hook=None) -> torch.Tensor:
        """Updates the statistics from ``rows2d`` [n, B*F] when accumulation is live
        (core/normalization.py:21-23, :30-37) and returns the device vector
        ``[mean | std]`` (:39-48) that the fused kernels consume.

        ``partial_sums_hook(sums_f64)`` may reduce the float64 sums across ranks in place
        before they are folded into the statistics (mesh-partitioned runs); it must also
        return the global row count."""
        sums = self.local_sums(rows2d, accumulate)
        n = 0
        if sums is not None:
            n = rows2d.shape[0]
            if partial_sums_hook is not None:
                n = int(partial_sums_hook(sums))
        return self.fold(sums, n)

    # ----------------------------------------------------------------- reference API
    def forward(self, batched_data, accumulate=True, partial_sums_hook=None):
        """Normalizes input data and accumulates statistics (core/normalization.py:19-24).
        ``partial_sums_hook`` as in :meth:`stats_for` (mesh-partitioned runs only)."""
        b, n, f = batched_data.shape
        rows = self._rows_view(batched_data.contiguous()).contiguous()
        ms = self.stats_for(rows, accumulate, partial_sums_hook)
        out = torch.empty_like(rows)
        ops.normalize(rows, ms, out, inverse=False)
        return out.reshape(n, b, f).permute(1, 0, 2).contiguous() if b > 1 else out.reshape(1, n, f)

    def inverse(self, normalized_batch_data):
        """Inverse transformation (core/normalization.py:26-28)."""
        b, n, f = normalized_batch_data.shape
        rows = self._rows_view(normalized_batch_data.contiguous()).contiguous()
        ms = self.stats_for(rows, accumulate=False)
        out = torch.empty_like(rows)
        ops.normalize(rows, ms, out, inverse=True)
        return out.reshape(n, b, f).permute(1, 0, 2).contiguous() if b > 1 else out.reshape(1, n, f)

    def _accumulate(self, batched_data, node_num=None):
        rows = self._rows_view(batched_data.contiguous()).contiguous()
        saved = self._max_accumulations
        self._max_accumulations = max(saved, self._num_accumulations + 1)   # unconditional, as in the reference
        try:
            self.stats_for(rows, accumulate=True)
        finally:
            self._max_accumulations = saved

    def _mean(self):
        ms = self.stats_for(None, accumulate=False)
        return ms[: self._acc_sum.numel()].reshape(self._acc_sum.shape).clone()

    def _std_with_epsilon(self):
        ms = self.stats_for(None, accumulate=False)
        return ms[self._acc_sum.numel():].reshape(self._acc_sum.shape).clone()

    def get_acc_sum(self):
        return self._acc_sum


# The reference pickles whole Normalizer modules (core/meshgraphnet.py:320-322), so the
# class has to resolve as ``core.normalization.Normalizer`` on both save and load.
Normalizer.__module__ = "core.normalization"


def ensure_core_alias():
    """Makes ``core.normalization.Normalizer`` importable as this class when no such
    module is present (i.e. outside the reference's own source tree)."""
    mod = sys.modules.get("core.normalization")
    if mod is not None and getattr(mod, "Normalizer", None) is Normalizer:
        return
    if mod is None:
        pkg = sys.modules.get("core")
        if pkg is None:
            pkg = types.ModuleType("core")
            pkg.__path__ = []           # namespace-like: lets ``import core.normalization`` resolve
            sys.modules["core"] = pkg
        mod = types.ModuleType("core.normalization")
        mod.Normalizer = Normalizer
        sys.modules["core.normalization"] = mod
        setattr(pkg, "normalization", mod)


def load_normalizer(path, device) -> Normalizer:
    """Reads a pickled normaliser without executing anything from the file: the tensor-only
    unpickler, with this class as the single extra allowed global."""
    with torch.serialization.safe_globals([(Normalizer, "core.normalization.Normalizer")]):
        obj = torch.load(path, map_location=device, weights_only=True)
    if not isinstance(obj, Normalizer):
        raise RuntimeError(f"{path} does not hold a Normalizer")
    return obj.to_device(device)
