"""Fused loss blocks on libgsx.so: value + analytic gradient in one pass over the render (csrc/loss.hip).

``fused_mapping_loss`` computes exactly what ``mapping_loss`` (gslam/backend.py:273-318) computes with ~60 torch
kernels: exposure affine, active-NeRF photometric term + 0.5 log^2(beta), 1 - fused_ssim('valid') on the un-exposed
rgb, the isotropic regulariser and the edge-aware depth TV, and hands autograd a ready gradient for the render tensor,
the exposure parameters and the log-scales.  ``fused_tracking_loss`` is the active-nerf tracking loss of
gslam/frontend.py:113-138,632-646.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from ._lib import check, lib, ptr, stream_ptr
from .ops import workspace
from .ssim import _strides


def loss_and_grads(render, alphas, gt, exposure, log_scales, vis_count, depth_index, beta_index, w_photo, w_ssim, w_iso,
                   w_tv, mode, backward_fn=None, iso_grad_fn=None, finish=True):
    """Raw fused block (no autograd): returns (out2, v_render, v_exposure, v_log_scales) where out2[0] = total =
    w_photo * photometric + w_ssim * (1 - ssim) + w_iso * isotropic + w_tv * tv and out2[1] = photometric, all on the
    device; the v_* are d total / d input.

    backward_fn(v_render): called as soon as the gradient of the render exists (the caller's render backward), before the
    isotropic term; iso_grad_fn() then returns the [N,3] gradient tensor the isotropic term is ADDED to in place (the
    returned v_log_scales is None in that case).  v_exposure is filled by the finishing launch at the end."""
    with torch.no_grad():
        render, gt = render.detach().contiguous(), gt.contiguous()
        exposure = exposure.detach()
        if log_scales is not None:
            log_scales = log_scales.detach()
        return _loss_and_grads(render, alphas, gt, exposure, log_scales, vis_count, depth_index, beta_index, w_photo,
                               w_ssim, w_iso, w_tv, mode, backward_fn, iso_grad_fn, finish)


def _loss_and_grads(render, alphas, gt, exposure, log_scales, vis_count, depth_index, beta_index, w_photo, w_ssim, w_iso,
                    w_tv, mode, backward_fn=None, iso_grad_fn=None, finish=True):
    """finish=False (photometric term only): the finishing launch is left to the caller - returns
    ((map_loss workspace, number of partial rows, loss coefficient), v_render, None, None)"""
    exposure = exposure.contiguous()
    Cn, H, W, CH = render.shape
    dev = render.device
    st = stream_ptr(dev)
    n_px = Cn * H * W
    v_render = torch.empty_like(render)
    v_exposure = torch.empty_like(exposure)
    ssim_grad = None
    ssim_ws, n_ssim = None, 0
    numel_ssim = Cn * 3 * (H - 10) * (W - 10)
    # every producer leaves its per-workgroup partial sums in its workspace; one launch finishes them all at the end
    if w_ssim != 0.0:
        # fused_ssim(outputs.rgbs NCHW-view, gt NCHW-view, 'valid') straight on the NHWC buffers (backend.py:303-307)
        s_r = (C.c_int64 * 4)(H * W * CH, 1, W * CH, CH)
        s_g = (C.c_int64 * 4)(H * W * 3, 1, W * 3, 3)
        dm = torch.empty(3, Cn, 3, H, W, dtype=torch.float32, device=dev)
        ssim_ws = workspace(lib.gsx_ssim_workspace_bytes(Cn, 3, H, W), dev, "ssim")
        n_ssim = lib.gsx_ssim_partials(Cn, 3, H, W)
        check(lib.gsx_ssim_fwd(ptr(render), ptr(gt), Cn, 3, H, W, s_r, s_g, 5, None, ptr(dm[0]), ptr(dm[1]),
                               ptr(dm[2]), ptr(ssim_ws), ssim_ws.numel(), st), "gsx_ssim_fwd")
        ssim_grad = torch.empty(Cn, 3, H, W, dtype=torch.float32, device=dev)
        one = _ones(dev)
        check(lib.gsx_ssim_bwd(ptr(render), ptr(gt), Cn, 3, H, W, s_r, s_g, 5, ptr(dm[0]), ptr(dm[1]), ptr(dm[2]),
                               ptr(one), -w_ssim / numel_ssim, ptr(ssim_grad), st), "gsx_ssim_bwd")
    denom = n_px * (3 if mode == 1 else 1)
    map_ws = workspace(lib.gsx_map_loss_workspace_bytes(Cn, H, W), dev, "map_loss")
    check(lib.gsx_map_loss(ptr(render), ptr(alphas.contiguous()) if alphas is not None else None, ptr(gt),
                           ptr(exposure), Cn, H, W, CH, depth_index, beta_index, mode, w_photo / denom, w_tv, 0.4,
                           ptr(ssim_grad), None, ptr(v_render), None, ptr(map_ws), map_ws.numel(), st),
          "gsx_map_loss")
    if backward_fn is not None:
        backward_fn(v_render)
    v_scales = None
    iso_ws, n_iso = None, 0
    if w_iso != 0.0 and log_scales is not None:
        log_scales = log_scales.contiguous()
        n_iso = log_scales.shape[0]
        iso_ws = workspace(lib.gsx_isotropic_workspace_bytes(n_iso), dev, "iso")
        into = iso_grad_fn() if iso_grad_fn is not None else None
        if into is not None:
            if not (into.is_contiguous() and into.shape == log_scales.shape and into.dtype == torch.float32):
                raise RuntimeError("iso_grad_fn must return a contiguous float32 [N,3] gradient")
            check(lib.gsx_isotropic_loss_acc(ptr(log_scales), ptr(vis_count.contiguous()), n_iso, w_iso, None,
                                             ptr(into), ptr(iso_ws), iso_ws.numel(), st), "gsx_isotropic_loss_acc")
        else:
            v_scales = torch.empty_like(log_scales)
            check(lib.gsx_isotropic_loss(ptr(log_scales), ptr(vis_count.contiguous()), n_iso, w_iso, None,
                                         ptr(v_scales), ptr(iso_ws), iso_ws.numel(), st), "gsx_isotropic_loss")
    pm = 1.0 / denom
    if not finish:
        if n_ssim or iso_ws is not None or w_tv != 0.0:
            raise ValueError("finish=False supports the photometric term alone")
        rows = Cn * ((H * W + 255) // 256)
        return (map_ws, rows, w_photo * pm), v_render, None, None
    # total / photometric from the raw sums (photometric, log-beta, tv, ssim, isotropic), on the device
    out2 = torch.empty(2, dtype=torch.float32, device=dev)
    c0 = (C.c_float * 5)(w_photo * pm, w_photo * pm, w_tv, -w_ssim / numel_ssim if n_ssim else 0.0,
                         w_iso if iso_ws is not None else 0.0)
    c1 = (C.c_float * 5)(pm, pm, 0.0, 0.0, 0.0)
    check(lib.gsx_loss_finish(ptr(map_ws), Cn, H, W, ptr(ssim_ws) if n_ssim else None, n_ssim,
                              ptr(iso_ws) if iso_ws is not None else None, n_iso, c0, c1, w_ssim if n_ssim else 0.0,
                              0.0, None, ptr(v_exposure), ptr(out2), st), "gsx_loss_finish")
    return out2, v_render, v_exposure, v_scales


class _FusedMappingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, alphas, gt, exposure, log_scales, vis_count, depth_index, beta_index, w_photo, w_ssim,
                w_iso, w_tv, mode):
        """returns (total, photometric) as 0-dim tensors"""
        out2, v_render, v_exposure, v_scales = loss_and_grads(render, alphas, gt, exposure, log_scales, vis_count,
                                                              depth_index, beta_index, w_photo, w_ssim, w_iso, w_tv,
                                                              mode)
        ctx.save_for_backward(v_render, v_exposure, v_scales)
        total, photo = out2[0].clone(), out2[1].clone()
        ctx.mark_non_differentiable(photo)
        return total, photo

    @staticmethod
    def backward(ctx, g_total, _g_photo):
        v_render, v_exposure, v_scales = ctx.saved_tensors
        return (v_render * g_total, None, None, v_exposure * g_total, None if v_scales is None else v_scales * g_total,
                None, None, None, None, None, None, None, None)


_ONES: dict = {}


def _ones(dev) -> Tensor:
    t = _ONES.get(dev)
    if t is None:
        t = torch.ones(1, dtype=torch.float32, device=dev)
        _ONES[dev] = t
    return t


def fused_mapping_loss(outputs, gt_imgs: Tensor, exposure_params: Tensor, log_scales: Tensor, *, ssim_weight: float,
                       iso_weight: float, tv_weight: float, active_gs: bool = True, shard: float = 1.0,
                       iso_scale: float = 1.0, vis_count: Optional[Tensor] = None):
    """(total, photometric) of gslam/backend.py:273-318 for a RasterizationOutput produced by
    gslam_amd.rasterization (needs its private un-split render).  ``shard`` = C_local / C_window for keyframe-sharded
    BA, ``iso_scale`` = 1 / world_size (SURVEY.md §8e loss-scaling rules)."""
    render = outputs._render
    if vis_count is None:
        vis_count = outputs._vis_count
    return _FusedMappingLoss.apply(render, outputs.alphas, gt_imgs, exposure_params, log_scales, vis_count,
                                   -1 if outputs._depth_index is None else outputs._depth_index,
                                   -1 if outputs._betas_index is None else outputs._betas_index,
                                   shard * (1.0 - ssim_weight), shard * ssim_weight, iso_scale * iso_weight, tv_weight,
                                   0 if active_gs else 1)


def mapping_loss_and_grads(outputs, gt_imgs: Tensor, exposure_params: Tensor, log_scales: Tensor, *, ssim_weight: float,
                           iso_weight: float, tv_weight: float, active_gs: bool = True, shard: float = 1.0,
                           iso_scale: float = 1.0, vis_count: Optional[Tensor] = None, backward_fn=None,
                           iso_grad_fn=None):
    """Same numbers as fused_mapping_loss without the autograd node: (out2[total, photometric], v_render, v_exposure,
    v_log_scales).  The caller seeds the backward with ``torch.autograd.backward([outputs._render], [v_render])``, or
    passes it as ``backward_fn`` (see loss_and_grads) to have the isotropic term added into the scale gradient in
    place."""
    if vis_count is None:
        vis_count = outputs._vis_count
    return loss_and_grads(outputs._render, outputs.alphas, gt_imgs, exposure_params, log_scales, vis_count,
                          -1 if outputs._depth_index is None else outputs._depth_index,
                          -1 if outputs._betas_index is None else outputs._betas_index,
                          shard * (1.0 - ssim_weight), shard * ssim_weight, iso_scale * iso_weight, tv_weight,
                          0 if active_gs else 1, backward_fn, iso_grad_fn)


def tracking_loss_and_grads(outputs, gt_img: Tensor, exposure_params: Tensor, finish: bool = True):
    """The tracking loss without the autograd node: (out2[loss, loss], v_render, v_exposure [2]); the caller seeds the
    backward with ``torch.autograd.backward([outputs._render], [v_render])`` (no multiply-by-one kernels, no clones).
    finish=False: ((partial rows workspace, n rows, loss coefficient), v_render, None) for gsx_track_opt_tail."""
    out2, v_render, v_exposure, _ = loss_and_grads(
        outputs._render, None, gt_img[None] if gt_img.dim() == 3 else gt_img, exposure_params.reshape(1, 2), None, None,
        -1 if outputs._depth_index is None else outputs._depth_index, outputs._betas_index, 1.0, 0.0, 0.0, 0.0, 2,
        finish=finish)
    return out2, v_render, (v_exposure.reshape(-1) if v_exposure is not None else None)


def fused_tracking_loss(outputs, gt_img: Tensor, exposure_params: Tensor):
    """active-nerf tracking loss of gslam/frontend.py:127 with the exposure affine of :632-636, C = 1."""
    render = outputs._render
    total, _ = _FusedMappingLoss.apply(render, None, gt_img[None] if gt_img.dim() == 3 else gt_img,
                                       exposure_params.reshape(1, 2), None, None,
                                       -1 if outputs._depth_index is None else outputs._depth_index,
                                       outputs._betas_index, 1.0, 0.0, 0.0, 0.0, 2)
    return total
