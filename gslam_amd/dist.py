"""Multi-GPU keyframe bundle adjustment: one process per GPU, ``torch.distributed`` over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).  New design - the reference is single-GPU with no collectives
(SURVEY.md §2.1, §8e).

Partitioning: the map (60 B/Gaussian params + 60 B grads + 120 B Adam state) is replicated; rank r renders the
keyframes {c : c mod G == r} of the BA window.  Per iteration there is exactly ONE data-path collective: an
all-reduce(sum) of a single contiguous fp32 bucket [N*15] that the six parameter ``.grad`` tensors are views of
(means3+quats4+scales3+opac1+colors3+log_unc1), plus one small int32 [N] all-reduce of per-Gaussian visible-camera
counts (isotropic term / opacity decay, backend.py:287,357).  xGMI is point-to-point (7 links x ~153 GB/s per GPU):
one large bucket lets RCCL drive all links; many small per-tensor all-reduces would be latency-bound.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as td

GRAD_PARAMS = ('means', 'quats', 'scales', 'opacities', 'colors', 'log_uncertainties')


def init_from_env(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract).
    Returns (rank, world_size); no-op (0, 1) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None and device.type == "cuda":
            kwargs["device_id"] = device
        td.init_process_group(backend=backend, **kwargs)
    return td.get_rank(), td.get_world_size()


class KeyframeShard:
    """Round-robin ownership of window keyframes by rank."""

    def __init__(self, group=None):
        self.group = group
        if td.is_available() and td.is_initialized():
            self.rank, self.world_size = td.get_rank(group), td.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1

    def select(self, window: Sequence) -> List:
        if self.world_size == 1:
            return list(window)
        return [f for i, f in enumerate(window) if i % self.world_size == self.rank]

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            td.all_reduce(t, op=td.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            td.all_reduce(t, op=td.ReduceOp.MAX, group=self.group)
        return t

    def broadcast_(self, tensors: Sequence[torch.Tensor], src: int = 0) -> None:
        if self.world_size > 1:
            for t in tensors:
                td.broadcast(t, src=src, group=self.group)


class GradBucket:
    """One flat fp32 buffer that the six splat ``.grad`` tensors are views of: autograd accumulates straight into
    it, a single all-reduce sums it across ranks and the fused Adam reads the views - no cat/split copies."""

    def __init__(self, splats, group=None):
        self.group = group
        self.splats = splats
        self._alloc()

    def _alloc(self):
        params = [getattr(self.splats, n) for n in GRAD_PARAMS]
        self._shapes = [tuple(p.shape) for p in params]
        total = sum(p.numel() for p in params)
        n = params[0].shape[0]
        # [N*15] gradients + [N] per-Gaussian visible-camera counts riding as a 16th fp32 column (exact: counts <= C)
        self.flat = torch.zeros(total + n, dtype=torch.float32, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        self.counts = self.flat[off:off + n]

    def attach_zeroed(self):
        params = [getattr(self.splats, n) for n in GRAD_PARAMS]
        if [tuple(p.shape) for p in params] != self._shapes:
            self._alloc()                                   # map was densified / pruned
        self.flat.zero_()
        for p, v in zip(params, self.views):
            p.grad = v

    def all_reduce(self):
        if td.is_available() and td.is_initialized() and td.get_world_size(self.group) > 1:
            td.all_reduce(self.flat, op=td.ReduceOp.SUM, group=self.group)
