"""Multi-tensor fused Adam on libgsx.so: one launch updates up to 8 parameter tensors with per-tensor learning
rates.  Semantics = torch.optim.Adam(fused=True) with the defaults the reference uses (betas (0.9, 0.999), eps 1e-8,
no weight decay, no amsgrad): gslam/backend.py:565-602 creates six such optimisers (one per splat attribute) plus a
pose optimiser; this class folds them into one kernel reading p, g, m, v once (28 B per element)."""
from __future__ import annotations

import ctypes as C
from typing import Iterable

import torch

from ._lib import check, lib, stream_ptr

_MAX = 8


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # bucket by (betas, eps, step) so that every launch shares its scalar state
        buckets: dict = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                    raise RuntimeError("FusedAdam needs contiguous float32 parameters on the GPU (no CPU fallback)")
                key = (group["betas"], group["eps"], st["step"], p.device)
                buckets.setdefault(key, []).append((p, p.grad.contiguous(), st, float(group["lr"])))
        for (betas, eps, step, dev), items in buckets.items():
            for i in range(0, len(items), _MAX):
                chunk = items[i:i + _MAX]
                n = len(chunk)
                arr = lambda xs: (C.c_void_p * n)(*xs)
                check(lib.gsx_adam_multi(
                    n, arr([p.data_ptr() for p, _, _, _ in chunk]), arr([g.data_ptr() for _, g, _, _ in chunk]),
                    arr([s["exp_avg"].data_ptr() for _, _, s, _ in chunk]),
                    arr([s["exp_avg_sq"].data_ptr() for _, _, s, _ in chunk]),
                    (C.c_int64 * n)(*[p.numel() for p, _, _, _ in chunk]),
                    (C.c_float * n)(*[lr for _, _, _, lr in chunk]),
                    float(betas[0]), float(betas[1]), float(eps), int(step), stream_ptr(dev)), "gsx_adam_multi")
        return loss
