"""Drop-in for gslam/rasterization.py: same ``rasterization(...)`` signature (pre-activation inputs) and the same
``RasterizationOutput`` dataclass, computed by the HIP kernels of libgsx.so.

Differences from the reference file are limited to what SURVEY.md §8a flags as dead or buggy there and which the
build must not copy: the ``ED`` / ``RGB+ED`` post-processing (KeyError 'depthaps', rasterization.py:342-344) is
implemented as the documented intent (depth / alpha), the >32-channel chunk loop (wrong variable at :323) and the
``covars`` path (:129-134 vs :147) are rejected explicitly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import ClassVar, Optional

import torch
from torch import Tensor

from . import ops
from .ops import PROJ_BETAS, PROJ_LOG_SCALES, PROJ_RENDER_DEPTH


@dataclass
class RasterizationOutput:
    """Field-for-field mirror of gslam/rasterization.py:17-41 (positional construction at backend.py:619 works)."""
    per_gaussian_params: ClassVar[tuple] = ('radii', 'means2d')
    rgbs: Tensor = None
    alphas: Tensor = None
    depthmaps: Tensor = None
    betas: Tensor = None
    tile_width: int = None
    tile_height: int = None
    tiles_per_gauss: Tensor = None
    isect_ids: Tensor = None
    flatten_ids: Tensor = None
    isect_offsets: Tensor = None
    width: int = None
    height: int = None
    tile_size: int = None
    n_cameras: int = None
    camera_ids: Tensor = None
    gaussian_ids: Tensor = None
    radii: Tensor = None
    means2d: Tensor = None
    depths: Tensor = None
    conics: Tensor = None
    opacities: Tensor = None
    n_touched: Tensor = None


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]
    log_scales: Tensor,  # [N, 3]
    logit_opacities: Tensor,  # [N]
    logit_colors: Tensor,  # [N, 3]
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    camera_model: str = "pinhole",
    covars: Optional[Tensor] = None,
    log_uncertainties: Optional[Tensor] = None,
    visibility_min_T: float = 0.5,
    mask: Optional[Tensor] = None,
) -> RasterizationOutput:
    """gslam ``rasterization`` (gslam/rasterization.py:44-360).  One fused projection+activation+packing kernel,
    tile intersection + depth sort, one tiled rasterisation kernel; autograd reaches every pre-activation input and
    ``viewmats``.  The only caller in the reference passes packed=False (map.py:99)."""
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert log_scales.shape == (N, 3), log_scales.shape
    assert logit_opacities.shape == (N,), logit_opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert render_mode in ["RGB", "D", "ED", "RGB+D", "RGB+ED"], render_mode
    if covars is not None:
        raise NotImplementedError("covars= is a dead branch in the reference (rasterization.py:129-134 vs :147)")
    if packed:
        raise NotImplementedError("packed=True is not used on the gslam hot path (map.py:99 passes packed=False); "
                                  "use gslam_amd.rasterization.get_new_splat_depth for the packed projection")
    if rasterize_mode != "classic" or camera_model != "pinhole" or sparse_grad:
        raise NotImplementedError("only rasterize_mode='classic', camera_model='pinhole', sparse_grad=False")
    if logit_colors.dim() != 2 or logit_colors.shape != (N, 3):
        raise NotImplementedError("logit_colors must be [N,3] (the reference never passes per-camera colours)")
    if render_mode in ("D", "ED"):
        raise NotImplementedError("depth-only modes are not used by gslam (map.py:83: 'RGB' or 'RGB+D')")
    if tile_size != 16:
        raise NotImplementedError("tile_size must be 16")

    flags = PROJ_LOG_SCALES
    ch = 3
    depth_index = betas_index = None
    if render_mode in ("RGB+D", "RGB+ED"):
        flags |= PROJ_RENDER_DEPTH
        depth_index = ch
        ch += 1
    if log_uncertainties is not None:
        flags |= PROJ_BETAS
        betas_index = ch
        ch += 1

    radii, means2d, depths, conics, _comps, rec, tiles_per_gauss, vis_count = ops._Projection.apply(
        means, quats, log_scales, viewmats, Ks, logit_opacities, logit_colors, log_uncertainties, int(width),
        int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip), False, flags, True, True, True)

    # backgrounds: [C,3] + 0 for depth + e^1 for beta (rasterization.py:236-239,251-255)
    bg = None
    if backgrounds is not None:
        parts = [backgrounds]
        if depth_index is not None:
            parts.append(torch.zeros(C, 1, device=backgrounds.device, dtype=backgrounds.dtype))
        if betas_index is not None:
            parts.append(torch.full((C, 1), math.e, device=backgrounds.device, dtype=backgrounds.dtype))
        bg = torch.cat(parts, dim=-1) if len(parts) > 1 else backgrounds

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    _, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height,
                                                packed=False, n_cameras=C, tiles_per_gauss=tiles_per_gauss)
    isect_offsets = ops.isect_offset_encode(isect_ids, C, tile_width, tile_height)

    render, alphas, n_touched, _last = ops._RasterizeRecords.apply(
        rec, means2d, conics, bg, isect_offsets, flatten_ids, ch, int(width), int(height), float(visibility_min_T),
        bool(absgrad))

    out = RasterizationOutput(
        rgbs=render[..., :3],
        alphas=alphas,
        tile_width=tile_width, tile_height=tile_height, tiles_per_gauss=tiles_per_gauss, isect_ids=isect_ids,
        flatten_ids=flatten_ids, isect_offsets=isect_offsets, width=width, height=height, tile_size=tile_size,
        n_cameras=C, camera_ids=None, gaussian_ids=None, radii=radii, means2d=means2d, depths=depths, conics=conics,
        opacities=rec[..., 5], n_touched=n_touched.long(),
    )
    if depth_index is not None:
        out.depthmaps = render[..., depth_index]
        if render_mode == "RGB+ED":
            out.depthmaps = out.depthmaps / alphas[..., 0].clamp(min=1e-10)
    if betas_index is not None:
        out.betas = render[..., betas_index]
    # private extras for the fused loss (gslam_amd.losses): the un-split render and per-Gaussian visibility counts
    out._render, out._depth_index, out._betas_index, out._vis_count = render, depth_index, betas_index, vis_count
    return out


def get_new_splat_depth(new_params: dict, viewmats: Tensor, Ks: Tensor, width: int, height: int,
                        near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                        eps2d: float = 0.3, packed: bool = True, sparse_grad: bool = False,
                        rasterize_mode: str = "classic", camera_model: str = "pinhole",
                        covars: Optional[Tensor] = None):
    """gslam/rasterization.py:363-448: projection only (used without grad by insertion.py:252-258)."""
    scales = torch.exp(new_params['scales'])
    res = ops.fully_fused_projection(new_params['means'], covars, new_params['quats'], scales, viewmats, Ks, width,
                                     height, eps2d=eps2d, packed=packed, near_plane=near_plane, far_plane=far_plane,
                                     radius_clip=radius_clip, sparse_grad=sparse_grad,
                                     calc_compensations=(rasterize_mode == "antialiased"), camera_model=camera_model)
    if packed:
        camera_ids, gaussian_ids, radii, means2d, depths, _, _ = res
        return camera_ids, gaussian_ids, radii, means2d, depths
    radii, means2d, depths, _, _ = res
    return radii, means2d, depths
