"""Drop-in for gslam/warp.py: ``Warp(K, H, W)(f1_pose, f2_pose, c1, d1) -> (result, normalized_warps, keep_mask)``.
One HIP kernel per direction; gradients reach both poses through T = f1 @ inv(f2) (gslam/warp.py:44)."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .ops import workspace


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, T, K, Kinv, c1, d1):
        if not T.is_cuda:
            raise _lib.GsxError("Warp runs on the GPU only (no CPU fallback)")
        T, K, Kinv = T.contiguous().float(), K.contiguous().float(), Kinv.contiguous().float()
        c1, d1 = c1.contiguous().float(), d1.contiguous().float()
        H, W = d1.shape
        dev = T.device
        result = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        nwarps = torch.empty(1, H, W, 2, dtype=torch.float32, device=dev)
        keep = torch.empty(H, W, dtype=torch.uint8, device=dev)
        check(lib.gsx_warp_fwd(ptr(T), ptr(K), ptr(Kinv), ptr(c1), ptr(d1), H, W, ptr(result), ptr(nwarps), ptr(keep),
                               stream_ptr(dev)), "gsx_warp_fwd")
        ctx.save_for_backward(T, K, Kinv, c1, d1)
        ctx.set_materialize_grads(False)
        keep = keep.bool()
        ctx.mark_non_differentiable(keep)
        return result, nwarps, keep

    @staticmethod
    def backward(ctx, v_result, v_nwarps, _v_keep):
        T, K, Kinv, c1, d1 = ctx.saved_tensors
        H, W = d1.shape
        dev = T.device
        if v_result is None and v_nwarps is None:
            return None, None, None, None, None
        v_result = torch.zeros(H, W, 3, dtype=torch.float32, device=dev) if v_result is None else v_result.contiguous()
        v_nwarps = None if v_nwarps is None else v_nwarps.contiguous()
        v_T = torch.empty(4, 4, dtype=torch.float32, device=dev)
        ws = workspace(lib.gsx_warp_bwd_workspace_bytes(H, W), dev, "warp")
        check(lib.gsx_warp_bwd(ptr(T), ptr(K), ptr(Kinv), ptr(c1), ptr(d1), H, W, ptr(v_result), ptr(v_nwarps),
                               ptr(v_T), ptr(ws), ws.numel(), stream_ptr(dev)), "gsx_warp_bwd")
        return v_T, None, None, None, None


class Warp(torch.nn.Module):
    def __init__(self, K: Tensor, H: int, W: int) -> None:
        super().__init__()
        self.H, self.W = H, W
        self.register_buffer("K", K)
        self.register_buffer("K_inv", torch.linalg.inv(K))        # gslam/warp.py:13

    def forward(self, f1_pose: Tensor, f2_pose: Tensor, c1: Tensor, d1: Tensor):
        """c1 [H,W,3], d1 [H,W].  c1/d1 are treated as constants (detached upstream: gslam/backend.py:513-514)."""
        T = f1_pose @ torch.linalg.inv(f2_pose)                     # gslam/warp.py:44
        return _WarpFn.apply(T, self.K, self.K_inv, c1.detach(), d1.detach())
