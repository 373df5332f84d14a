"""``fused_ssim(img1, img2, padding, train)`` with the contract of rahul-goel/fused-ssim as used at
gslam/backend.py:303-307: images [B,CH,H,W] (any strides - the reference passes NHWC->NCHW permutes), returns the
scalar mean SSIM, differentiable w.r.t. ``img1`` only."""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .ops import workspace


def _strides(t: Tensor):
    return (C.c_int64 * 4)(*t.stride())


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, crop, train):
        if not img1.is_cuda:
            raise _lib.GsxError("fused_ssim runs on the GPU only (no CPU fallback)")
        assert img1.dtype == torch.float32 and img2.dtype == torch.float32 and img1.shape == img2.shape
        B, CH, H, W = img1.shape
        dev = img1.device
        out = torch.empty(1, dtype=torch.float32, device=dev)
        need = train and img1.requires_grad
        dm = torch.empty(3, B, CH, H, W, dtype=torch.float32, device=dev) if need else None
        ws = workspace(lib.gsx_ssim_workspace_bytes(B, CH, H, W), dev, "ssim")
        check(lib.gsx_ssim_fwd(ptr(img1), ptr(img2), B, CH, H, W, _strides(img1), _strides(img2), crop, ptr(out),
                               ptr(dm[0]) if need else None, ptr(dm[1]) if need else None,
                               ptr(dm[2]) if need else None, ptr(ws), ws.numel(), stream_ptr(dev)), "gsx_ssim_fwd")
        numel = B * CH * (H - 2 * crop) * (W - 2 * crop)
        ctx.save_for_backward(img1, img2, dm)
        ctx.cfg = (crop, numel)
        return (out / numel).reshape(())

    @staticmethod
    def backward(ctx, v_out):
        img1, img2, dm = ctx.saved_tensors
        crop, numel = ctx.cfg
        if dm is None:
            raise RuntimeError("fused_ssim was called with train=False; no backward state was kept")
        B, CH, H, W = img1.shape
        dev = img1.device
        g = torch.empty(B, CH, H, W, dtype=torch.float32, device=dev)
        scale = v_out.reshape(1).to(torch.float32).contiguous()
        check(lib.gsx_ssim_bwd(ptr(img1), ptr(img2), B, CH, H, W, _strides(img1), _strides(img2), crop, ptr(dm[0]),
                               ptr(dm[1]), ptr(dm[2]), ptr(scale), 1.0 / numel, ptr(g), stream_ptr(dev)),
              "gsx_ssim_bwd")
        return g, None, None, None


def fused_ssim(img1: Tensor, img2: Tensor, padding: str = "same", train: bool = True) -> Tensor:
    assert padding in ("same", "valid"), padding
    return _FusedSSIM.apply(img1, img2, 5 if padding == "valid" else 0, bool(train))
