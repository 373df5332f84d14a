"""``gsplat.rendering.rasterization``-compatible entry point (post-activation inputs), as called by the reference at
pipeline.py:106-116,122-132: ``rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height)``
-> ``(render_colors[C,H,W,D], render_alphas[C,H,W,1], meta)``.  Defaults follow upstream gsplat 1.4.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops


def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor, viewmats: Tensor, Ks: Tensor,
    width: int, height: int, near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
    eps2d: float = 0.3, sh_degree: Optional[int] = None, packed: bool = True, tile_size: int = 16,
    backgrounds: Optional[Tensor] = None, render_mode: str = "RGB", sparse_grad: bool = False, absgrad: bool = False,
    rasterize_mode: str = "classic", channel_chunk: int = 32, distributed: bool = False,
    camera_model: str = "pinhole", covars: Optional[Tensor] = None, visibility_min_T: float = 0.5,
) -> Tuple[Tensor, Tensor, Dict]:
    """Inputs are post-activation.  ``packed`` only changes upstream's memory layout, not the result; this build
    always computes the dense [C,N] layout and reports it in ``meta`` (camera_ids/gaussian_ids are None)."""
    N, C = means.shape[0], viewmats.shape[0]
    assert render_mode in ["RGB", "D", "ED", "RGB+D", "RGB+ED"], render_mode
    if distributed:
        raise NotImplementedError("use gslam_amd.dist for multi-GPU bundle adjustment")
    antialiased = rasterize_mode == "antialiased"
    radii, means2d, depths, conics, comps = ops.fully_fused_projection(
        means, covars, quats, scales, viewmats, Ks, width, height, eps2d=eps2d, packed=False,
        near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, sparse_grad=sparse_grad,
        calc_compensations=antialiased, camera_model=camera_model)
    opac = opacities.unsqueeze(0).expand(C, -1)
    if comps is not None:
        opac = opac * comps

    if sh_degree is None:
        cols = colors if colors.dim() == 3 else colors.unsqueeze(0).expand(C, -1, -1)
    else:
        campos = torch.inverse(viewmats)[:, :3, 3]                        # [C,3]
        dirs = means[None, :, :] - campos[:, None, :]                     # [C,N,3]
        coeffs = colors if colors.dim() == 3 else None
        if coeffs is None:
            raise ValueError("sh_degree given: colors must be SH coefficients [N,K,3]")
        cols = ops.spherical_harmonics(sh_degree, dirs, coeffs, masks=radii)

    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat((cols, depths[..., None]), dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=backgrounds.device)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, device=backgrounds.device)

    tile_width, tile_height = math.ceil(width / float(tile_size)), math.ceil(height / float(tile_size))
    tiles_per_gauss, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, tile_size, tile_width,
                                                              tile_height, packed=False, n_cameras=C)
    isect_offsets = ops.isect_offset_encode(isect_ids, C, tile_width, tile_height)
    render_colors, render_alphas, n_touched = ops.rasterize_to_pixels(
        means2d, conics, cols.contiguous(), opac.contiguous(), width, height, tile_size, isect_offsets, flatten_ids,
        backgrounds=backgrounds, packed=False, absgrad=absgrad, visibility_min_T=visibility_min_T)
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat([render_colors[..., :-1],
                                   render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
    meta = {
        "camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths,
        "conics": conics, "opacities": opac, "tile_width": tile_width, "tile_height": tile_height,
        "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets, "width": width, "height": height, "tile_size": tile_size, "n_cameras": C,
        "n_touched": n_touched,
    }
    return render_colors, render_alphas, meta
