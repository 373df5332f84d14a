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
    camera_model: str = "pinhole", covars: Optional[Tensor] = None, visibility_min_T: float = 0.5, capacity=None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Inputs are post-activation.  ``packed`` only changes upstream's memory layout, not the result; this build
    always computes the dense [C,N] layout and reports it in ``meta`` (camera_ids/gaussian_ids are None).
    ``capacity`` (extension): a gslam_amd.rasterization.IsectCapacity makes the render sync-free - the tile lists go into
    capacity-sized buffers without the read-back of the intersection count, so the call can sit inside a captured HIP graph;
    ``meta['flatten_ids']`` is then the capacity-sized buffer, ``meta['isect_ids']`` None and ``meta['n_isects']`` the
    device-side count (poll ``capacity.validate()`` for overflow)."""
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
        # camera centre = inv(viewmat)[:3, 3] = -A^-1 t of the affine [A | t]; the 3x3 inverse is written out with cross
        # products (rows r0, r1, r2: A^-1 = [r1 x r2, r2 x r0, r0 x r1] / det) - torch.inverse goes through a solver
        # library that synchronises, which a captured HIP graph does not permit
        A, tv = viewmats[:, :3, :3], viewmats[:, :3, 3]
        c0 = torch.cross(A[:, 1], A[:, 2], dim=-1)
        c1 = torch.cross(A[:, 2], A[:, 0], dim=-1)
        c2 = torch.cross(A[:, 0], A[:, 1], dim=-1)
        det = (A[:, 0] * c0).sum(-1, keepdim=True)
        campos = -(c0 * tv[:, 0:1] + c1 * tv[:, 1:2] + c2 * tv[:, 2:3]) / det         # [C,3]
        coeffs = colors if colors.dim() == 3 else None
        if coeffs is None:
            raise ValueError("sh_degree given: colors must be SH coefficients [N,K,3]")
        # view directions means - campos are formed inside the SH kernels (no [C,N,3] array: 120 MB written and read back
        # at 5 M Gaussians); gradients reach the means directly and the view matrices through campos
        cols = ops.spherical_harmonics_from_means(sh_degree, means, campos, coeffs, masks=radii)

    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat((cols, depths[..., None]), dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=backgrounds.device)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, device=backgrounds.device)

    tile_width, tile_height = math.ceil(width / float(tile_size)), math.ceil(height / float(tile_size))
    n_isects = None
    if capacity is None:
        tiles_per_gauss, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, tile_size, tile_width,
                                                                  tile_height, packed=False, n_cameras=C)
        isect_offsets = ops.isect_offset_encode(isect_ids, C, tile_width, tile_height)
        render_colors, render_alphas, n_touched = ops.rasterize_to_pixels(
            means2d, conics, cols.contiguous(), opac.contiguous(), width, height, tile_size, isect_offsets, flatten_ids,
            backgrounds=backgrounds, packed=False, absgrad=absgrad, visibility_min_T=visibility_min_T)
    else:
        if tile_size != 16:
            raise NotImplementedError("tile_size must be 16")
        dev = means.device
        with torch.no_grad():
            tiles_per_gauss = torch.empty(C, N, dtype=torch.int32, device=dev)
            from ._lib import check, lib, ptr, stream_ptr
            check(lib.gsx_isect_count(ptr(means2d.detach()), ptr(radii), C * N, tile_width, tile_height,
                                      ptr(tiles_per_gauss), stream_ptr(dev)), "gsx_isect_count")
            if capacity.capacity == 0:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("render this shape once eagerly before capturing (intersection capacity probe)")
                capacity.ensure(int(tiles_per_gauss.sum().item()))
            cap = capacity.capacity
            flatten_ids = torch.empty(cap, dtype=torch.int32, device=dev)
            off1, n_isects, _ = ops.isect_bin_sort(means2d.detach(), radii, depths.detach(), tile_width, tile_height, cap,
                                                   None, flatten_ids, status=capacity.status)
            capacity.M_dev = n_isects
        isect_ids = None
        isect_offsets = off1[:-1].view(C, tile_height, tile_width)
        render_colors, render_alphas, n_touched = ops.rasterize_to_pixels(
            means2d, conics, cols.contiguous(), opac.contiguous(), width, height, tile_size, off1, flatten_ids,
            backgrounds=backgrounds, packed=False, absgrad=absgrad, visibility_min_T=visibility_min_T,
            offsets_have_end=True)
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat([render_colors[..., :-1],
                                   render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
    meta = {
        "camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths,
        "conics": conics, "opacities": opac, "tile_width": tile_width, "tile_height": tile_height,
        "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets, "width": width, "height": height, "tile_size": tile_size, "n_cameras": C,
        "n_touched": n_touched, "n_isects": n_isects,
    }
    return render_colors, render_alphas, meta
