"""Drop-in for gslam/rasterization.py: same ``rasterization(...)`` signature (pre-activation inputs) and the same
``RasterizationOutput`` fields, computed by the HIP kernels of libgsx.so.

Differences from the reference file are limited to what SURVEY.md §8a flags as dead or buggy there and which the
build must not copy: the ``ED`` / ``RGB+ED`` post-processing (KeyError 'depthaps', rasterization.py:342-344) is
implemented as the documented intent (depth / alpha), the >32-channel chunk loop (wrong variable at :323) and the
``covars`` path (:129-134 vs :147) are rejected explicitly.

The live argument set (map.py:88-103: [N,3] colours, 'classic', 'RGB' / 'RGB+D') runs on the fused kernels; the depth-only
modes ride on them; everything else the signature admits - per-camera colours, other colour widths, 'antialiased' - is
composed from the operator-level entry points (``_rasterization_composed``).

Sizes: by default M is read back once per render, like the reference's ``isect_tiles`` (exact arrays).  With
``capacity=IsectCapacity(...)`` the render is sync-free: the tile-binned sort keeps every size on the device and writes
into capacity-sized buffers; ``isect_ids`` / ``flatten_ids`` are trimmed lazily (first access) and overflow is read from
the sticky device status word (``IsectCapacity.validate()``).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from . import ops
from .ops import PROJ_BETAS, PROJ_LOG_SCALES, PROJ_RENDER_DEPTH


class RasterizationOutput:
    """Field-for-field mirror of the dataclass at gslam/rasterization.py:17-41 (same field order, so the positional
    construction ``RasterizationOutput(None, alphas, depthmaps)`` of backend.py:619 works).  ``isect_ids`` and
    ``flatten_ids`` are materialised on first access when they come from the sync-free path."""
    per_gaussian_params = ('radii', 'means2d')
    _FIELDS = ('rgbs', 'alphas', 'depthmaps', 'betas', 'tile_width', 'tile_height', 'tiles_per_gauss', 'isect_ids',
               'flatten_ids', 'isect_offsets', 'width', 'height', 'tile_size', 'n_cameras', 'camera_ids',
               'gaussian_ids', 'radii', 'means2d', 'depths', 'conics', 'opacities', 'n_touched')

    def __init__(self, rgbs=None, alphas=None, depthmaps=None, betas=None, tile_width=None, tile_height=None,
                 tiles_per_gauss=None, isect_ids=None, flatten_ids=None, isect_offsets=None, width=None, height=None,
                 tile_size=None, n_cameras=None, camera_ids=None, gaussian_ids=None, radii=None, means2d=None,
                 depths=None, conics=None, opacities=None, n_touched=None):
        self.rgbs, self.alphas, self.depthmaps, self.betas = rgbs, alphas, depthmaps, betas
        self.tile_width, self.tile_height, self.tiles_per_gauss = tile_width, tile_height, tiles_per_gauss
        self._isect_ids, self._flatten_ids, self.isect_offsets = isect_ids, flatten_ids, isect_offsets
        self.width, self.height, self.tile_size, self.n_cameras = width, height, tile_size, n_cameras
        self.camera_ids, self.gaussian_ids, self.radii, self.means2d = camera_ids, gaussian_ids, radii, means2d
        self.depths, self.conics, self.opacities, self._n_touched = depths, conics, opacities, n_touched
        self._lazy = None      # (_IsectBuffers, generation) when the id arrays live in capacity-sized buffers
        self._render = self._depth_index = self._betas_index = self._vis_count = None

    def _materialise(self):
        if self._lazy is not None:
            bufs, M_dev, flat, ids_needed = self._lazy
            M = int(M_dev.item())                      # the one read-back, only if somebody asks for the arrays
            if M > flat.shape[0]:
                raise RuntimeError(f"isect capacity overflow: {M} intersections > capacity {flat.shape[0]}; "
                                   "call IsectCapacity.validate() and re-render")
            self._flatten_ids = flat[:M]
            self._isect_ids = ops.rebuild_isect_ids(self, M)
            self._lazy = None

    @property
    def n_touched(self):
        """int64 like the reference's ``n_touched.long()`` (rasterization.py:352); converted on first access"""
        if self._n_touched is not None and self._n_touched.dtype != torch.int64:
            self._n_touched = self._n_touched.long()
        return self._n_touched

    @n_touched.setter
    def n_touched(self, v):
        self._n_touched = v

    @property
    def flatten_ids(self):
        self._materialise()
        return self._flatten_ids

    @property
    def isect_ids(self):
        self._materialise()
        return self._isect_ids

    def __repr__(self):
        return "RasterizationOutput(" + ", ".join(f"{f}=..." for f in self._FIELDS) + ")"


class IsectCapacity:
    """Caller-owned capacity of the tile lists for SYNC-FREE eager renders (``rasterization(..., capacity=cap)``).

    The reference reads the intersection count M back to the host inside ``isect_tiles`` on every render; by default this
    module does the same (exact sizes, one read-back).  A caller that renders the same shape over and over - a loop under
    HIP-graph capture, a benchmark - passes one of these instead: the first render probes M once (synchronously), later
    renders write into buffers of 1.5x that size without any read-back, and ``validate()`` tells - with one blocking read
    of the sticky device status word - whether any render since the last call overflowed (the capacity has then been
    grown; the caller re-renders).  One object per problem shape (N, C, W, H) and stream of launches; no global state.
    The optimisation loops do not come through here: their launch plans (gslam_amd.plan) own their capacity."""
    GROW = 1.5

    def __init__(self, device):
        self.dev = torch.device(device)
        self.capacity = 0
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.M_dev: Optional[Tensor] = None
        self.last_M = 0

    def ensure(self, estimate: int):
        want = int(estimate * self.GROW) + 4096
        if want > self.capacity:
            self.capacity = want

    def validate(self) -> bool:
        ok = True
        if self.M_dev is not None:
            st, m = int(self.status.item()), int(self.M_dev.item())
            self.last_M = m
            if st & 2:                         # hardening flag of the binning kernels: lists built from corrupt counts
                self.status.zero_()
                raise RuntimeError(f"corrupt tile counts in a sync-free render (status {st}, M {m}, capacity "
                                   f"{self.capacity})")
            if st & 1:
                ok = False
                # M can be an under-estimate when the overflow was in the pre-sort's instance records (csrc/isect_bin.hip
                # 3c): grow geometrically from the current capacity as well
                self.capacity = int(max(m, self.capacity) * self.GROW) + 4096
                self.status.zero_()
            else:
                self.ensure(m)
        return ok


_BG_CACHE: "dict" = {}


def _packed_backgrounds(backgrounds: Optional[Tensor], C: int, with_depth: bool, with_beta: bool) -> Optional[Tensor]:
    """[C,3] -> [C,CH]: + 0 for depth + e^1 for beta (rasterization.py:236-239,251-255).  Cached while the caller
    keeps passing the same (unmodified) tensor, which is what map.py does every render; a small FIFO."""
    if backgrounds is None:
        return None
    if not (with_depth or with_beta):
        return backgrounds
    key = (backgrounds.data_ptr(), backgrounds._version, C, with_depth, with_beta)
    hit = _BG_CACHE.get(key)
    if hit is not None and not backgrounds.requires_grad and hit[0] is backgrounds:
        return hit[1]
    parts = [backgrounds]
    if with_depth:
        parts.append(torch.zeros(C, 1, device=backgrounds.device, dtype=backgrounds.dtype))
    if with_beta:
        parts.append(torch.full((C, 1), math.e, device=backgrounds.device, dtype=backgrounds.dtype))
    bg = torch.cat(parts, dim=-1)
    if not backgrounds.requires_grad:
        while len(_BG_CACHE) >= 16:
            _BG_CACHE.pop(next(iter(_BG_CACHE)))
        _BG_CACHE[key] = (backgrounds, bg)          # holding the source pins its address while the entry lives
    return bg


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
    need_n_touched: bool = True,
    capacity: Optional[IsectCapacity] = None,
) -> RasterizationOutput:
    """gslam ``rasterization`` (gslam/rasterization.py:44-360).  One fused projection+activation+packing kernel,
    tile-binned depth sort, one tiled rasterisation kernel; autograd reaches every pre-activation input and
    ``viewmats``.  The only caller in the reference passes packed=False (map.py:99).

    ``need_n_touched`` (extension, default True = reference behaviour): False skips the per-Gaussian touched-pixel
    counts, which only visibility pruning reads (backend.py:370-375); ``n_touched`` is then None.
    ``capacity`` (extension): an ``IsectCapacity`` makes the render sync-free (see there); default None sizes the tile
    lists exactly from one read-back of M, like the reference's ``isect_tiles``."""
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
    if camera_model != "pinhole" or sparse_grad:
        raise NotImplementedError("only camera_model='pinhole', sparse_grad=False")
    assert rasterize_mode in ("classic", "antialiased"), rasterize_mode
    assert (logit_colors.dim() == 2 and logit_colors.shape[0] == N) or (
        logit_colors.dim() == 3 and logit_colors.shape[:2] == (C, N)), logit_colors.shape       # rasterization.py:141-143
    if rasterize_mode != "classic" or logit_colors.dim() != 2 or logit_colors.shape != (N, 3):
        # argument sets no caller in gslam uses (map.py:88-103 passes [N,3] colours, 'classic'): per-camera colours, colour
        # widths other than 3, antialiased opacity compensation - composed from the operators, not the fused kernels
        if tile_size != 16:
            raise NotImplementedError("tile_size must be 16")
        return _rasterization_composed(means, quats, log_scales, logit_opacities, logit_colors, viewmats, Ks, width, height,
                                       near_plane, far_plane, radius_clip, eps2d, packed, backgrounds, render_mode, absgrad,
                                       rasterize_mode, log_uncertainties, visibility_min_T)
    # depth-only modes (rasterization.py:242-246; no caller in gslam, map.py:83 passes 'RGB' or 'RGB+D'): the depth channel of
    # the RGB+D render IS the 'D' render - every channel is composited with the same weights - so these modes go through
    # the RGB+D kernels, return ``rgbs=None`` and a zero colour background (:244-245) keeps the unused channels inert
    depth_only = render_mode in ("D", "ED")
    if depth_only and backgrounds is not None:
        backgrounds = torch.zeros(C, 3, dtype=torch.float32, device=means.device)
    if tile_size != 16:
        raise NotImplementedError("tile_size must be 16")

    flags = PROJ_LOG_SCALES
    ch = 3
    depth_index = betas_index = None
    if render_mode in ("RGB+D", "RGB+ED", "D", "ED"):
        flags |= PROJ_RENDER_DEPTH
        depth_index = ch
        ch += 1
    if log_uncertainties is not None:
        flags |= PROJ_BETAS
        betas_index = ch
        ch += 1

    # the backward's gradient records: cleared by the projection kernel on its way (no zero-fill launch per backward)
    needs_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (means, quats, log_scales, viewmats, logit_opacities, logit_colors,
                                                    log_uncertainties))
    v_rec_buf = torch.empty(C, N, 12, dtype=torch.float32, device=means.device) if needs_grad else None
    # frozen map (tracking) and no depth channel: the colour / opacity columns of the gradient records feed nothing,
    # the rasteriser backward then reduces five values per survivor instead of 6 + CH
    geom_only = needs_grad and depth_index is None and not any(
        t is not None and t.requires_grad for t in (logit_opacities, logit_colors, log_uncertainties))
    radii, means2d, depths, conics, _comps, rec, tiles_per_gauss, vis_count = ops._Projection.apply(
        means, quats, log_scales, viewmats, Ks, logit_opacities, logit_colors, log_uncertainties, int(width),
        int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip), False, flags, True, True,
        True, v_rec_buf)

    packed_means2d = packed_sel = None
    if packed:
        # packed=True: the [nnz, 2] means2d the caller gets must be the node the rasteriser's gradient passes through
        # (means2d.retain_grad() on the packed array, as with gsplat's packed projection): pack first, and hand the
        # rasteriser the dense array rebuilt from the packed rows (same values: culled rows are zero either way)
        packed_sel = torch.nonzero((radii > 0).reshape(-1)).squeeze(1)
        packed_means2d = means2d.reshape(-1, 2)[packed_sel]
        means2d = torch.zeros(C * N, 2, dtype=means2d.dtype, device=means2d.device).index_put(
            (packed_sel,), packed_means2d).view(C, N, 2)

    # backgrounds: [C,3] + 0 for depth + e^1 for beta (rasterization.py:236-239,251-255)
    bg = _packed_backgrounds(backgrounds, C, depth_index is not None, betas_index is not None)

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    dev = means.device

    lazy = None
    tile_order = None
    if capacity is None:
        # reference-shaped: M is read back inside isect_tiles (one host sync per render, exact sizes)
        _, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height,
                                                    packed=False, n_cameras=C, tiles_per_gauss=tiles_per_gauss)
        isect_offsets = ops.isect_offset_encode(isect_ids, C, tile_width, tile_height)
        raster_offsets, has_end = isect_offsets, False
    else:
        pool = capacity
        capturing = torch.cuda.is_current_stream_capturing()
        if pool.capacity == 0:
            # first render of this shape: size the buffers with ONE synchronous probe of the projection's upper bound
            if capturing:
                raise RuntimeError("render this (N, C, W, H) once eagerly before capturing a HIP graph "
                                   "(intersection capacity probe)")
            pool.ensure(int(tiles_per_gauss.sum().item()))
        cap = pool.capacity
        flat_buf = torch.empty(cap, dtype=torch.int32, device=dev)
        # heaviest-first launch order for the rasteriser: pays off while the tile lists are short (100 k Gaussians:
        # -5..7 % rasteriser time at 1 and 8 cameras); with ~1500 entries per tile every tile saturates and the spatial
        # order is as good or better (tools/ab_raster.py)
        n_t = C * tile_height * tile_width
        tile_order = torch.empty(n_t, dtype=torch.int32, device=dev) if cap < 1000 * n_t else None
        with torch.no_grad():
            off1, M_dev, _ = ops.isect_bin_sort(means2d.detach(), radii, depths.detach(), tile_width, tile_height, cap,
                                                None, flat_buf, status=pool.status, tile_order=tile_order)
        pool.M_dev = M_dev
        isect_offsets = off1[:-1].view(C, tile_height, tile_width)
        raster_offsets, has_end, flatten_ids, isect_ids = off1, True, flat_buf, None
        lazy = (pool, M_dev, flat_buf, True)

    render, alphas, n_touched, _last = ops._RasterizeRecords.apply(
        rec, means2d, conics, bg, raster_offsets, flatten_ids, ch, int(width), int(height), float(visibility_min_T),
        bool(absgrad), has_end, bool(need_n_touched), v_rec_buf, tile_order, geom_only)

    out = RasterizationOutput(
        rgbs=None if depth_only else render[..., :3],
        alphas=alphas,
        tile_width=tile_width, tile_height=tile_height, tiles_per_gauss=tiles_per_gauss, isect_ids=isect_ids,
        flatten_ids=flatten_ids, isect_offsets=isect_offsets, width=width, height=height, tile_size=tile_size,
        n_cameras=C, camera_ids=None, gaussian_ids=None, radii=radii, means2d=means2d, depths=depths, conics=conics,
        opacities=rec[..., 5], n_touched=n_touched,
    )
    out._lazy = lazy
    if depth_index is not None:
        out.depthmaps = render[..., depth_index]
        if render_mode in ("RGB+ED", "ED"):
            out.depthmaps = out.depthmaps / alphas[..., 0].clamp(min=1e-10)
    if betas_index is not None:
        out.betas = render[..., betas_index]
    # private extras for the fused loss (gslam_amd.losses): the un-split render and per-Gaussian visibility counts
    out._render, out._depth_index, out._betas_index, out._vis_count = render, depth_index, betas_index, vis_count
    if packed:
        _pack_output(out, N, C, packed_sel, packed_means2d)
    return out


def _rasterization_composed(means, quats, log_scales, logit_opacities, logit_colors, viewmats, Ks, width, height, near_plane,
                            far_plane, radius_clip, eps2d, packed, backgrounds, render_mode, absgrad, rasterize_mode,
                            log_uncertainties, visibility_min_T) -> RasterizationOutput:
    """gslam/rasterization.py:145-360 step by step over the operator-level entry points (torch activations, projection,
    channel packing, binning, rasteriser with its 5-channel chunks): every argument set the signature admits and the fused
    kernels do not specialise for.  Same outputs, same autograd contract (``means2d`` is a graph node)."""
    N, C = means.shape[0], viewmats.shape[0]
    opac = torch.sigmoid(logit_opacities)                                           # :145-149
    colors = torch.sigmoid(logit_colors)
    scales = torch.exp(log_scales)
    betas = torch.exp(log_uncertainties).clamp(min=0.01) if log_uncertainties is not None else None
    radii, means2d, depths, conics, comps = ops.fully_fused_projection(
        means, None, quats, scales, viewmats, Ks, int(width), int(height), eps2d=eps2d, packed=False,
        near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip,
        calc_compensations=(rasterize_mode == "antialiased"))                       # :153-170
    opacities = opac.repeat(C, 1)                                                   # :187
    if comps is not None:
        opacities = opacities * comps                                               # :190-191
    packed_means2d = packed_sel = None
    if packed:
        packed_sel = torch.nonzero((radii > 0).reshape(-1)).squeeze(1)
        packed_means2d = means2d.reshape(-1, 2)[packed_sel]
        means2d = torch.zeros(C * N, 2, dtype=means2d.dtype, device=means2d.device).index_put(
            (packed_sel,), packed_means2d).view(C, N, 2)
    cols = colors if colors.dim() == 3 else colors.unsqueeze(0).expand(C, -1, -1)    # :222-228
    depth_only = render_mode in ("D", "ED")
    depth_index = betas_index = None
    if render_mode in ("RGB+D", "RGB+ED"):                                          # :234-240
        cols = torch.cat((cols, depths[..., None]), dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=backgrounds.device)], dim=-1)
        depth_index = cols.shape[-1] - 1
    elif depth_only:                                                                # :241-246
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, device=backgrounds.device)
        depth_index = 0
    if betas is not None:                                                           # :249-256
        cols = torch.cat((cols, betas[None, :, None].expand(C, -1, -1)), dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.full((C, 1), math.e, device=backgrounds.device)], dim=-1)
        betas_index = cols.shape[-1] - 1
    tile_width, tile_height = math.ceil(width / 16.0), math.ceil(height / 16.0)
    tiles_per_gauss, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, 16, tile_width, tile_height,
                                                              packed=False, n_cameras=C)
    isect_offsets = ops.isect_offset_encode(isect_ids, C, tile_width, tile_height)
    render, alphas, n_touched = ops.rasterize_to_pixels(means2d, conics, cols.contiguous(), opacities.contiguous(),
                                                        int(width), int(height), 16, isect_offsets, flatten_ids,
                                                        backgrounds=backgrounds, packed=False, absgrad=absgrad,
                                                        visibility_min_T=visibility_min_T)
    out = RasterizationOutput(
        rgbs=None if depth_only else render[..., :3],                               # :347-348 keeps the first three channels
        alphas=alphas,
        tile_width=tile_width, tile_height=tile_height, tiles_per_gauss=tiles_per_gauss, isect_ids=isect_ids,
        flatten_ids=flatten_ids, isect_offsets=isect_offsets, width=width, height=height, tile_size=16,
        n_cameras=C, camera_ids=None, gaussian_ids=None, radii=radii, means2d=means2d, depths=depths, conics=conics,
        opacities=opacities, n_touched=n_touched,
    )
    out._lazy = None
    if depth_index is not None:
        out.depthmaps = render[..., depth_index]
        if render_mode in ("RGB+ED", "ED"):
            out.depthmaps = out.depthmaps / alphas[..., 0].clamp(min=1e-10)
    if betas_index is not None:
        out.betas = render[..., betas_index]
    out._render, out._depth_index, out._betas_index = render, depth_index, betas_index
    out._vis_count = (radii > 0).sum(0).to(torch.int32)
    if packed:
        _pack_output(out, N, C, packed_sel, packed_means2d)
    return out


def _pack_output(out: RasterizationOutput, N: int, C: int, sel: Tensor, means2d_packed: Tensor) -> None:
    """packed=True (the signature's default, gslam/rasterization.py:58,174-182): the images are those of the dense render -
    packing is a layout of the per-(camera, Gaussian) arrays - and the per-pair fields become [nnz, ...] over the visible
    pairs in flatten-id order with ``camera_ids`` / ``gaussian_ids`` beside them, as gsplat's packed projection returns
    them; ``flatten_ids`` then index those packed rows.  The row selection is differentiable indexing, so ``means2d`` is
    still a graph node that supports ``retain_grad()`` (backend.py:326).  One host sync (nnz), as in the reference's packed
    path; the only caller in the reference passes packed=False (map.py:99)."""
    vis = out.radii > 0                                                   # sel: flatten ids of the visible pairs, ascending
    out.camera_ids, out.gaussian_ids = sel // N, sel % N
    rank = torch.cumsum(vis.reshape(-1).to(torch.int64), 0) - 1           # flatten id -> packed row
    take = lambda t, tail=(): t.reshape((C * N,) + tuple(tail))[sel]
    out.radii = take(out.radii)
    out.means2d = means2d_packed
    out.depths = take(out.depths)
    out.conics = take(out.conics, (3,))
    out.opacities = take(out.opacities)
    out.tiles_per_gauss = take(out.tiles_per_gauss)
    if out.n_touched is not None:
        out.n_touched = take(out.n_touched)
    out._flatten_ids = rank[out.flatten_ids.long()].to(torch.int32)       # (the property materialises the sync-free buffers)


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
