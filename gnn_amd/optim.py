"""Adam on the HIP path: one kernel launch for all parameter tensors.

Drop-in for the ``torch.optim.Adam(model.parameters(), lr=..., weight_decay=...)`` of
run_train_multiscale.py:115-119 (same update rule, ``param_groups`` and ``state`` layout, so
``CosineAnnealingLR`` (:122-126) and ``state_dict()`` work unchanged); ``amsgrad``, ``maximize`` and
sparse gradients are not supported.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

CHUNK = 1 << 14          # elements per workgroup


class FusedAdam(torch.optim.Optimizer):
    """``capturable=True``: the scalars of the update (learning rate, bias corrections) travel through a pinned host
    record and a device record instead of kernel arguments, and the pointer table is uploaded with an asynchronous
    copy, so that ``step()`` can be captured in a HIP graph (``gnn_amd.training.CapturedTrainStep``); before every
    replay :meth:`advance_host` does the host part of a step.  Results are the same bit for bit."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = bool(capturable)

    def _write_hyper(self, gi, group, step):
        """Fills the pinned host record of group ``gi`` for step number ``step``.  The asynchronous copy of the
        previous step reads the same record when the GPU gets to it, not when it was enqueued: wait for it first."""
        rec = self.__dict__.setdefault("_hyper", {}).get(gi)
        if rec is None:
            host = torch.zeros(8, dtype=torch.float32).pin_memory()
            rec = self._hyper[gi] = [host, torch.zeros(8, dtype=torch.float32, device=group["params"][0].device), None]
        if rec[2] is not None:
            if not torch.cuda.is_current_stream_capturing():     # (a capture starts from a synchronised device)
                rec[2].synchronize()
            rec[2] = None
        rc = _lib.lib().mgn_adam_hyper(float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                       float(group["weight_decay"]), int(step), C.c_void_p(rec[0].data_ptr()))
        _lib.check(rc, "mgn_adam_hyper")
        return rec

    def advance_host(self):
        """The host part of a step whose device part is replayed from a captured graph: step counts, and the update's
        scalars (from the current ``lr`` of every group) into the pinned records the graph copies from.  The caller
        makes sure the previous replay has finished (its copy reads the same records)."""
        for gi, group in enumerate(self.param_groups):
            hit = self.__dict__.get("_fast", {}).get(gi)
            if hit is None:
                continue
            for st in hit[1]:
                st["step"] += 1
            self._write_hyper(gi, group, hit[1][0]["step"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        fast = self.__dict__.setdefault("_fast", {})
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            dev = params[0].device
            # the per-parameter state dictionaries, looked up once (hashing a tensor costs more than the rest of this
            # loop); load_state_dict / __setstate__ drop the list, and the identity check catches a replaced entry list
            hit = fast.get(gi)
            if hit is None or len(hit[0]) != len(params) or any(a is not b for a, b in zip(hit[0], params)):
                states = []
                for p in params:
                    st = self.state[p]
                    if st and torch.is_tensor(st["step"]):       # a state_dict written by torch.optim.Adam keeps the count as a tensor
                        st["step"] = int(st["step"].item())
                    if not st:
                        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                            raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors on the GPU")
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    states.append(st)
                hit = fast[gi] = (list(params), states)
            states = hit[1]
            for st in states:
                st["step"] += 1
            step = states[0]["step"]
            # the pointer table is rebuilt and uploaded only when an address changed (gradients are
            # usually re-allocated at the same addresses step after step)
            # (load_state_dict / a replaced state entry installs new moment tensors: their addresses are part of the key)
            key = (tuple(p.grad.data_ptr() for p in params) + tuple(p.data_ptr() for p in params)
                   + tuple(st["exp_avg"].data_ptr() for st in states) + tuple(st["exp_avg_sq"].data_ptr() for st in states))
            cache = self.__dict__.setdefault("_tables", {})
            hit = cache.get(gi)
            if hit is None or hit[0] != key:
                rows, keep = [], []
                for p, st in zip(params, states):
                    g = p.grad
                    if g.is_sparse or g.dtype != torch.float32 or g.device != p.device or p.device != dev:
                        raise RuntimeError("FusedAdam: gradients must be dense fp32 tensors on the parameters' device")
                    if st["step"] != step:
                        raise RuntimeError("FusedAdam: all parameters of a group must share their step count")
                    if not g.is_contiguous():
                        g = g.contiguous()
                        keep.append(g)
                    n = p.numel()
                    for off in range(0, n, CHUNK):
                        b = 4 * off
                        rows.append((p.data_ptr() + b, g.data_ptr() + b, st["exp_avg"].data_ptr() + b,
                                     st["exp_avg_sq"].data_ptr() + b, min(CHUNK, n - off)))
                host_rows = torch.from_numpy(np.asarray(rows, dtype=np.int64))
                if self.capturable:
                    # asynchronous upload from a pinned buffer that is kept (allocating pinned memory is not legal under
                    # stream capture, an asynchronous copy from it is); the previous copy may not have run yet: wait
                    rec = self._write_hyper(gi, group, step)            # creates the record, waits for its event
                    pin = self.__dict__.setdefault("_pinned_rows", {}).get(gi)
                    if pin is None or pin.shape != host_rows.shape:
                        pin = self._pinned_rows[gi] = torch.empty(host_rows.shape, dtype=torch.int64).pin_memory()
                    pin.copy_(host_rows)
                    host_rows = pin
                    table = torch.empty(host_rows.shape, dtype=torch.int64, device=dev)
                    table.copy_(host_rows, non_blocking=True)
                else:
                    table = host_rows.to(dev, non_blocking=False)
                # copies of strided gradients are not reusable; the pinned rows live as long as the table (a captured copy reads them)
                hit = (key if not keep else None, table, len(rows), keep, host_rows)
                cache[gi] = hit
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                if self.capturable:
                    rec = self._write_hyper(gi, group, step)
                    rec[1].copy_(rec[0], non_blocking=True)
                    rc = _lib.lib().mgn_adam_step_dev(C.c_void_p(hit[1].data_ptr()), hit[2], C.c_void_p(rec[1].data_ptr()), stream)
                    _lib.check(rc, "mgn_adam_step_dev")
                    if not torch.cuda.is_current_stream_capturing():
                        rec[2] = torch.cuda.Event()
                        rec[2].record()
                else:
                    rc = _lib.lib().mgn_adam_step(C.c_void_p(hit[1].data_ptr()), hit[2], float(group["lr"]),
                                                  float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                                  float(group["weight_decay"]), int(step), stream)
                    _lib.check(rc, "mgn_adam_step")
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__["_fast"] = {}                  # new state dictionaries

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__["_fast"] = {}
