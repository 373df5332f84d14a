"""Low-level operators with the call signatures of ``gsplat.cuda._wrapper`` as used by the reference
(gslam/rasterization.py:9-14,153-170,261-274,325-339; gslam/insertion.py:88), implemented as
``torch.autograd.Function`` wrappers that hand raw device pointers + the current HIP stream to libgsx.so.

PyTorch is plumbing here (device memory, streams, autograd graph); all arithmetic is in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

TILE = 16
PROJ_LOG_SCALES, PROJ_RENDER_DEPTH, PROJ_BETAS = 1, 2, 4


# ---------------------------------------------------------------------------------------------------------------
# workspace cache: one growing byte buffer per (device, tag); reuse is safe because all work is stream ordered
# ---------------------------------------------------------------------------------------------------------------
_ws: dict = {}


def workspace(nbytes: int, device, tag: str = "default") -> Tensor:
    """scratch of the eager operators (the launch plans of gslam_amd.plan own theirs): an outgrown buffer is simply
    dropped - the caching allocator keeps it alive until the launches already queued on its stream have run"""
    key = (torch.device(device).index, tag, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.5), 1 << 16), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _f32c(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_cuda:
        raise _lib.GsxError(f"{name}: gslam_amd ops run on the GPU only (no CPU fallback)")
    return t.contiguous()


def record_stride(ch: int) -> int:
    rs = lib.gsx_record_stride(ch)
    if rs < 0:
        raise ValueError(f"unsupported channel count {ch}")
    return rs


# ---------------------------------------------------------------------------------------------------------------
# K1 / K2
# ---------------------------------------------------------------------------------------------------------------
class _Projection(torch.autograd.Function):
    """gsplat fully_fused_projection (unpacked).  Optionally fuses the gslam front-end (log-scales, sigmoid
    opacities/colours, betas, record packing; gslam/rasterization.py:145-149,183-256)."""

    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, logit_opac, logit_colors, log_unc, width, height, eps2d,
                near_plane, far_plane, radius_clip, calc_comp, flags, want_rec, want_tiles, want_vis=False,
                v_rec_clear=None):
        """v_rec_clear: an (uninitialised) [C,N,12] buffer the kernel clears on the way: the rasteriser's backward
        accumulates its gradient records into it (pass the same tensor to _RasterizeRecords), which saves the separate
        zero-fill launch in front of every backward."""
        means, quats, scales = _f32c(means, "means"), _f32c(quats, "quats"), _f32c(scales, "scales")
        # a caller that consumes the pose gradient itself (tracking.GraphedTracker's fused closure tail) marks its view
        # matrices: the backward then leaves the per-workgroup partials in the "proj_bwd" workspace and returns no grad
        ctx.view_partials = bool(getattr(viewmats, "_gsx_partials_only", False))
        # viewmats that come straight out of primitives.pose_batch carry a link object: the backward then leaves its
        # pose partials to the PoseZhou backward (one launch instead of finishing pass + pose backward).  Only the
        # first projection that consumes a given viewmats tensor takes the link; any other one returns v_viewmats.
        link = getattr(viewmats, "_gsx_pose_link", None)
        if link is not None and not ctx.view_partials and not link.claimed and viewmats.shape[0] == link.count:
            link.claimed = True
            ctx.pose_link = link
        else:
            ctx.pose_link = None
        viewmats, Ks = _f32c(viewmats, "viewmats"), _f32c(Ks, "Ks")
        N, Cn = means.shape[0], viewmats.shape[0]
        dev = means.device
        radii = torch.empty(Cn, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(Cn, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(Cn, N, dtype=torch.float32, device=dev)
        conics = torch.empty(Cn, N, 3, dtype=torch.float32, device=dev)
        comps = torch.empty(Cn, N, dtype=torch.float32, device=dev) if calc_comp else None
        tile_w, tile_h = math.ceil(width / TILE), math.ceil(height / TILE)
        tiles = torch.empty(Cn, N, dtype=torch.int32, device=dev) if want_tiles else None
        vis = torch.empty(N, dtype=torch.int32, device=dev) if want_vis else None
        rec = None
        if want_rec:
            logit_opac, logit_colors = _f32c(logit_opac, "logit_opacities"), _f32c(logit_colors, "logit_colors")
            if log_unc is not None:
                log_unc = _f32c(log_unc, "log_uncertainties")
            rec = torch.empty(Cn, N, 12, dtype=torch.float32, device=dev)
        check(lib.gsx_project_fwd(ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), N, Cn, width, height,
                                  eps2d, near_plane, far_plane, radius_clip, flags, ptr(radii), ptr(means2d),
                                  ptr(depths), ptr(conics), ptr(comps), ptr(tiles), tile_w, tile_h, ptr(logit_opac),
                                  ptr(logit_colors), ptr(log_unc), ptr(rec), ptr(vis),
                                  ptr(v_rec_clear) if want_rec else None, stream_ptr(dev)), "gsx_project_fwd")
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii,
                              logit_opac if want_rec else None, logit_colors if want_rec else None,
                              log_unc if want_rec else None)
        ctx.cfg = (width, height, eps2d, near_plane, far_plane, flags, want_rec, calc_comp)
        ctx.set_materialize_grads(False)        # undefined output grads arrive as None, not as zero-filled tensors
        ctx.mark_non_differentiable(radii)
        if tiles is not None:
            ctx.mark_non_differentiable(tiles)
        if vis is not None:
            ctx.mark_non_differentiable(vis)
        return radii, means2d, depths, conics, comps, rec, tiles, vis

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps, v_rec, _v_tiles, _v_vis):
        if v_means2d is None and v_depths is None and v_conics is None and v_comps is None and v_rec is None:
            return (None,) * 20
        means, quats, scales, viewmats, Ks, radii, logit_opac, logit_colors, log_unc = ctx.saved_tensors
        width, height, eps2d, near_plane, far_plane, flags, want_rec, calc_comp = ctx.cfg
        N, Cn = means.shape[0], viewmats.shape[0]
        dev = means.device

        def strided(v, width_):
            """(tensor kept alive, row stride in floats) for a [C,N,width_] gradient that may be a view of v_rec"""
            if v is None:
                return torch.zeros(Cn, N, width_, dtype=torch.float32, device=dev), width_
            ok = (v.stride(-1) == 1 and v.stride(0) == N * v.stride(1)) if v.dim() == 3 else False
            if not ok:
                v = v.contiguous()
            return v, v.stride(1)

        v_means2d, s_m2d = strided(v_means2d, 2)
        v_conics, s_con = strided(v_conics, 3)
        v_depths = None if v_depths is None else v_depths.contiguous()
        v_comps = None if (v_comps is None or not calc_comp) else v_comps.contiguous()
        if want_rec and v_rec is None:
            v_rec = torch.zeros(Cn, N, 12, dtype=torch.float32, device=dev)
        if v_rec is not None:
            v_rec = v_rec.contiguous()
        need_view = ctx.needs_input_grad[3]
        link = ctx.pose_link if need_view else None
        partials_only = need_view and (ctx.view_partials or link is not None)
        if partials_only:
            flags |= 8                                          # GSX_PROJ_VIEW_PARTIALS
        # tracking (frozen map): only the pose gradient is wanted; the kernel then skips the per-Gaussian chain and stores
        need_gauss = any(ctx.needs_input_grad[i] for i in (0, 1, 2, 5, 6, 7)) or not need_view
        v_means = torch.empty_like(means) if need_gauss else None
        v_quats = torch.empty_like(quats) if need_gauss else None
        v_scales = torch.empty_like(scales) if need_gauss else None
        v_view = torch.empty(Cn, 4, 4, dtype=torch.float32, device=dev) if (need_view and not partials_only) else None
        v_lo = v_lc = v_lu = None
        if want_rec and need_gauss:
            v_lo = torch.empty_like(logit_opac)
            v_lc = torch.empty_like(logit_colors)
            v_lu = torch.empty_like(log_unc) if log_unc is not None else None
        ws_bytes = lib.gsx_project_bwd_workspace_bytes(N, Cn)
        if link is not None:        # the partials outlive this call: their own buffer, not the shared workspace
            ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
            # tagged with the id of this backward pass: a pose backward of ANOTHER pass must not pick them up (a pass that
            # never reaches the pose nodes - autograd.grad with restricted inputs - would otherwise leave them behind)
            link.partials = (ws, int(lib.gsx_project_bwd_blocks(N)), torch._C._current_graph_task_id())
        else:
            ws = workspace(ws_bytes, dev, "proj_bwd")
        check(lib.gsx_project_bwd(ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), N, Cn, width, height,
                                  eps2d, near_plane, far_plane, flags, ptr(radii), ptr(v_means2d), s_m2d,
                                  ptr(v_depths), ptr(v_conics), s_con, ptr(v_comps), ptr(logit_opac),
                                  ptr(logit_colors), ptr(log_unc), ptr(v_rec) if want_rec else None, ptr(v_means),
                                  ptr(v_quats), ptr(v_scales), ptr(v_view), ptr(v_lo), ptr(v_lc), ptr(v_lu), ptr(ws),
                                  ws.numel(), stream_ptr(dev)), "gsx_project_bwd")
        return (v_means, v_quats, v_scales, v_view, None, v_lo, v_lc, v_lu) + (None,) * 12


def fully_fused_projection(means: Tensor, covars: Optional[Tensor], quats: Optional[Tensor], scales: Optional[Tensor],
                           viewmats: Tensor, Ks: Tensor, width: int, height: int, eps2d: float = 0.3,
                           near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                           packed: bool = False, sparse_grad: bool = False, calc_compensations: bool = False,
                           camera_model: str = "pinhole"):
    """Same contract as gsplat's op at gslam/rasterization.py:153-170 / :390-407.

    unpacked -> (radii[C,N] i32, means2d[C,N,2], depths[C,N], conics[C,N,3], compensations|None)
    packed   -> (camera_ids, gaussian_ids, radii[nnz], means2d[nnz,2], depths[nnz], conics[nnz,3], compensations)
    """
    if covars is not None:
        raise NotImplementedError("covars= is a dead branch in the reference (rasterization.py:129-134 vs :147)")
    if camera_model != "pinhole":
        raise NotImplementedError("only the pinhole camera model is on the gslam hot path")
    if sparse_grad:
        raise NotImplementedError("sparse_grad is never enabled by the reference (rasterization.py:62)")
    radii, means2d, depths, conics, comps, _, _, _ = _Projection.apply(
        means, quats, scales, viewmats, Ks, None, None, None, int(width), int(height), float(eps2d),
        float(near_plane), float(far_plane), float(radius_clip), bool(calc_compensations), 0, False, False)
    if not packed:
        return radii, means2d, depths, conics, comps
    cam_ids, gauss_ids = torch.nonzero(radii > 0, as_tuple=True)   # row-major = ascending flatten id
    out = (cam_ids, gauss_ids, radii[cam_ids, gauss_ids], means2d[cam_ids, gauss_ids], depths[cam_ids, gauss_ids],
           conics[cam_ids, gauss_ids], None if comps is None else comps[cam_ids, gauss_ids])
    return out


def quat_scale_to_covar_preci(quats: Tensor, scales: Tensor, compute_covar: bool = True, compute_preci: bool = True,
                              triu: bool = False):
    """gsplat.quat_scale_to_covar_preci as used (under no_grad) at gslam/insertion.py:88-91."""
    quats, scales = _f32c(quats, "quats"), _f32c(scales, "scales")
    n = quats.shape[0]
    covars = torch.empty(n, 3, 3, dtype=torch.float32, device=quats.device)
    precis = torch.empty(n, 3, 3, dtype=torch.float32, device=quats.device) if compute_preci else None
    check(lib.gsx_quat_scale_to_covar_preci(ptr(quats.detach()), ptr(scales.detach()), n, ptr(covars), ptr(precis),
                                            stream_ptr(quats.device)), "gsx_quat_scale_to_covar_preci")
    if triu:
        # upper-triangular 6-vectors (xx, xy, xz, yy, yz, zz), the order of gslam/rasterization.py:133-134; unused by gslam
        iu = ([0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2])
        covars = covars[..., iu[0], iu[1]]
        precis = None if precis is None else precis[..., iu[0], iu[1]]
    return (covars if compute_covar else None), precis


# ---------------------------------------------------------------------------------------------------------------
# K3..K7
# ---------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int, tile_height: int,
                sort: bool = True, packed: bool = False, n_cameras: Optional[int] = None,
                camera_ids: Optional[Tensor] = None, gaussian_ids: Optional[Tensor] = None,
                tiles_per_gauss: Optional[Tensor] = None):
    """gsplat isect_tiles (gslam/rasterization.py:261-272) -> (tiles_per_gauss[C,N] i32, isect_ids[M] i64,
    flatten_ids[M] i32).  One device->host read-back of M, like the reference."""
    if packed:
        return _isect_tiles_packed(means2d, radii, depths, tile_size, tile_width, tile_height, sort, n_cameras,
                                   camera_ids, gaussian_ids)
    if tile_size != TILE:
        raise NotImplementedError("tile_size must be 16 (the only value the reference uses, rasterization.py:59)")
    means2d, depths = _f32c(means2d.detach(), "means2d"), _f32c(depths.detach(), "depths")
    radii = radii.contiguous()
    assert radii.dtype == torch.int32
    Cn, N = radii.shape
    dev = means2d.device
    st = stream_ptr(dev)
    if tiles_per_gauss is None:
        tiles_per_gauss = torch.empty(Cn, N, dtype=torch.int32, device=dev)
        check(lib.gsx_isect_count(ptr(means2d), ptr(radii), Cn * N, tile_width, tile_height, ptr(tiles_per_gauss), st),
              "gsx_isect_count")
    if sort:
        # tile-binned sort (csrc/isect_bin.hip).  This gsplat-shaped API returns exactly-sized arrays, so M is read
        # back once here; the fused gslam path (gslam_amd.rasterization with an IsectCapacity, gslam_amd.plan) uses the
        # same kernels without any read-back.
        M = int(tiles_per_gauss.sum().item()) if Cn * N > 0 else 0
        isect_ids = torch.empty(M, dtype=torch.int64, device=dev)
        flatten_ids = torch.empty(M, dtype=torch.int32, device=dev)
        if M > 0:
            off, _, _ = isect_bin_sort(means2d, radii, depths, tile_width, tile_height, M, isect_ids, flatten_ids)
            # the binning already produced the per-tile offsets: isect_offset_encode(isect_ids, ...) hands them out
            # instead of re-deriving them from the 64-bit keys (175 us for 76 M intersections)
            isect_ids._gsx_offsets = (off[:-1].view(Cn, tile_height, tile_width), isect_ids._version)
        return tiles_per_gauss, isect_ids, flatten_ids
    # sort=False: gsplat's unsorted emission order (flatten order, tiles row-major): own scan + emit
    cum = torch.empty(Cn * N, dtype=torch.int64, device=dev)
    M = 0
    if Cn * N > 0:
        ws = workspace(lib.gsx_scan_workspace_bytes(Cn * N), dev, "scan")
        check(lib.gsx_isect_scan(ptr(tiles_per_gauss), Cn * N, ptr(cum), ptr(ws), ws.numel(), st), "gsx_isect_scan")
        out = C.c_int64(0)
        check(lib.gsx_read_i64(cum[-1:].data_ptr(), C.byref(out), st), "gsx_read_i64")
        M = int(out.value)
    isect_ids = torch.empty(M, dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(M, dtype=torch.int32, device=dev)
    if M > 0:
        check(lib.gsx_isect_emit(ptr(means2d), ptr(radii), ptr(depths), ptr(cum), N, Cn, tile_width, tile_height, M,
                                 ptr(isect_ids), ptr(flatten_ids), st), "gsx_isect_emit")
    return tiles_per_gauss, isect_ids, flatten_ids


@torch.no_grad()
def _isect_tiles_packed(means2d, radii, depths, tile_size, tile_width, tile_height, sort, n_cameras, camera_ids,
                        gaussian_ids):
    """packed=True (gslam/rasterization.py:261-272 with the [nnz] arrays of the packed projection, :174-182): per-row camera
    ids instead of a [C,N] layout; ``flatten_ids`` index the packed rows.  Not on gslam's live path (map.py:99 passes
    packed=False): served by the dense kernels over a [C, max rows per camera] layout that keeps every camera's rows in
    their packed order - the sort key (camera, tile, depth bits) and the tie order (ascending row) are the same, so the
    outputs equal gsplat's packed ones entry for entry.  One extra read-back (rows per camera)."""
    if n_cameras is None or camera_ids is None or gaussian_ids is None:
        raise ValueError("packed isect_tiles needs n_cameras, camera_ids and gaussian_ids")
    nnz = int(radii.shape[0])
    assert means2d.shape == (nnz, 2) and depths.shape == (nnz,) and camera_ids.shape == (nnz,)
    dev = means2d.device
    Cn = int(n_cameras)
    if nnz == 0:
        e = lambda dt: torch.empty(0, dtype=dt, device=dev)
        return e(torch.int32), e(torch.int64), e(torch.int32)
    cam = camera_ids.long()
    cnt = torch.bincount(cam, minlength=Cn)
    start = torch.cumsum(cnt, 0) - cnt
    order = torch.argsort(cam, stable=True)                    # rows grouped by camera, packed order kept inside a camera
    j = torch.empty(nnz, dtype=torch.int64, device=dev)
    j[order] = torch.arange(nnz, device=dev) - start[cam[order]]
    Np = int(cnt.max().item())
    dense = lambda t, tail=(): torch.zeros((Cn, Np) + tuple(tail), dtype=t.dtype, device=dev).index_put((cam, j), t)
    tpg_d, isect_ids, flat_d = isect_tiles(dense(_f32c(means2d.detach(), "means2d"), (2,)), dense(radii.to(torch.int32)),
                                           dense(_f32c(depths.detach(), "depths")), tile_size, tile_width, tile_height,
                                           sort=sort, n_cameras=Cn)
    row_of = torch.full((Cn * Np,), -1, dtype=torch.int64, device=dev)
    row_of[cam * Np + j] = torch.arange(nnz, device=dev)
    flatten_ids = row_of[flat_d.long()].to(torch.int32)
    return tpg_d[cam, j].contiguous(), isect_ids, flatten_ids


@torch.no_grad()
def isect_bin_sort(means2d: Tensor, radii: Tensor, depths: Tensor, tile_width: int, tile_height: int, capacity: int,
                   isect_ids: Optional[Tensor], flatten_ids: Tensor, offsets: Optional[Tensor] = None,
                   M_dev: Optional[Tensor] = None, status: Optional[Tensor] = None,
                   tile_order: Optional[Tensor] = None):
    """Sync-free K3..K7: fills ``flatten_ids`` (and ``isect_ids``) up to ``capacity`` and returns
    (offsets int32 [T+1], M_dev int64 [1], status int32 [1]); status bit 0 = capacity overflow.  ``tile_order``
    (int32 [T], optional) receives the heaviest-first launch order for the rasteriser."""
    Cn, N = radii.shape
    dev = means2d.device
    T = Cn * tile_width * tile_height
    offsets = torch.empty(T + 1, dtype=torch.int32, device=dev) if offsets is None else offsets
    M_dev = torch.empty(1, dtype=torch.int64, device=dev) if M_dev is None else M_dev
    status = torch.zeros(1, dtype=torch.int32, device=dev) if status is None else status
    ws = workspace(lib.gsx_isect_bin_workspace_bytes_n(Cn, N, tile_width, tile_height, capacity), dev, "isect_bin")
    check(lib.gsx_isect_bin_sort(ptr(means2d), ptr(radii), ptr(depths), N, Cn, tile_width, tile_height, capacity,
                                 ptr(offsets), ptr(M_dev), ptr(status), ptr(isect_ids), ptr(flatten_ids),
                                 ptr(tile_order), ptr(ws), ws.numel(), stream_ptr(dev)), "gsx_isect_bin_sort")
    return offsets, M_dev, status


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """gsplat isect_offset_encode (gslam/rasterization.py:274) -> int32 [C, tile_h, tile_w]."""
    cached = getattr(isect_ids, "_gsx_offsets", None)
    if cached is not None and cached[1] == isect_ids._version and \
            tuple(cached[0].shape) == (n_cameras, tile_height, tile_width):
        return cached[0]
    isect_ids = isect_ids.contiguous()
    dev = isect_ids.device
    offsets = torch.empty(n_cameras, tile_height, tile_width, dtype=torch.int32, device=dev)
    check(lib.gsx_isect_offset_encode(ptr(isect_ids), isect_ids.shape[0], n_cameras, tile_width, tile_height,
                                      ptr(offsets), stream_ptr(dev)), "gsx_isect_offset_encode")
    return offsets


# ---------------------------------------------------------------------------------------------------------------
# K8 / K9
# ---------------------------------------------------------------------------------------------------------------
class _RasterizeRecords(torch.autograd.Function):
    """Rasterise splat records.  ``means2d`` and ``conics`` are graph pass-throughs: the kernel reads them from the
    record, but their gradients are returned on these inputs so that ``means2d.retain_grad()`` works as in the
    reference (gslam/backend.py:326, insertion.py:298).  The xy/conic columns of the returned v_rec alias them and
    are ignored by the projection backward."""

    @staticmethod
    def forward(ctx, rec, means2d, conics, backgrounds, offsets, flatten_ids, ch, width, height, vis_min_T, absgrad,
                has_end=False, want_touched=True, v_rec_buf=None, tile_order=None, geometry_only=False):
        """geometry_only: the caller only needs the xy / conic gradient columns (frozen map, no depth channel).
        tile_order: int32 [T] launch order of the tiles (isect_bin_sort), a scheduling hint.
        v_rec_buf: a [C,N,RS] buffer already cleared by the projection forward (see _Projection) for the backward's
        gradient records; used once, a second backward through the same node allocates its own.
        has_end: ``offsets`` is the flat int32 [T+1] array of gsx_isect_bin_sort and ``flatten_ids`` a
        capacity-sized buffer (sync-free path); otherwise the gsplat layout ([C,tile_h,tile_w] offsets, exact M).
        want_touched=False skips the per-Gaussian touched-pixel counts (n_touched comes back as None)."""
        rec = _f32c(rec, "rec")
        Cn, N, RS = rec.shape
        assert RS == record_stride(ch)
        dev = rec.device
        tile_w, tile_h = math.ceil(width / TILE), math.ceil(height / TILE)
        assert offsets.numel() == Cn * tile_h * tile_w + (1 if has_end else 0)
        bg = None if backgrounds is None else _f32c(backgrounds, "backgrounds")
        render = torch.empty(Cn, height, width, ch, dtype=torch.float32, device=dev)
        alphas = torch.empty(Cn, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(Cn, height, width, dtype=torch.int32, device=dev)
        n_touched = torch.zeros(Cn, N, dtype=torch.int32, device=dev) if want_touched else None
        offsets, flatten_ids = offsets.contiguous(), flatten_ids.contiguous()
        M = flatten_ids.shape[0]
        check(lib.gsx_raster_fwd(ptr(rec), ch, ptr(bg), ptr(offsets), ptr(flatten_ids), M, 1 if has_end else 0, Cn,
                                 width, height, tile_w, tile_h, vis_min_T, ptr(render), ptr(alphas), ptr(last_ids),
                                 ptr(n_touched), ptr(tile_order), stream_ptr(dev)), "gsx_raster_fwd")
        ctx.save_for_backward(rec, bg, offsets, flatten_ids, alphas, last_ids)
        ctx.tile_order = tile_order
        ctx.geometry_only = bool(geometry_only)
        ctx.set_materialize_grads(False)
        ctx.cfg = (ch, width, height, absgrad, has_end)
        ctx.means2d_ref = means2d
        ctx.v_rec_buf = v_rec_buf if (v_rec_buf is not None and tuple(v_rec_buf.shape) == (Cn, N, RS)) else None
        if n_touched is not None:
            ctx.mark_non_differentiable(n_touched, last_ids)
        else:
            ctx.mark_non_differentiable(last_ids)
        return render, alphas, n_touched, last_ids

    @staticmethod
    def backward(ctx, v_render, v_alphas, _v_nt, _v_last):
        rec, bg, offsets, flatten_ids, alphas, last_ids = ctx.saved_tensors
        ch, width, height, absgrad, has_end = ctx.cfg
        Cn, N, RS = rec.shape
        dev = rec.device
        tile_w, tile_h = math.ceil(width / TILE), math.ceil(height / TILE)
        if v_render is None and v_alphas is None:
            return (None,) * 16
        v_render = torch.zeros_like(alphas).expand(-1, -1, -1, ch).contiguous() if v_render is None \
            else v_render.contiguous()
        v_alphas = None if v_alphas is None else v_alphas.contiguous()      # NULL = zero gradient (kernel-side)
        v_rec, ctx.v_rec_buf = ctx.v_rec_buf, None
        if v_rec is None:
            v_rec = torch.zeros(Cn, N, RS, dtype=torch.float32, device=dev)
        v_abs = torch.zeros(Cn, N, 2, dtype=torch.float32, device=dev) if absgrad else None
        check(lib.gsx_raster_bwd(ptr(rec), ch, ptr(bg), ptr(offsets), ptr(flatten_ids), flatten_ids.shape[0],
                                 1 if has_end else 0, Cn, width, height, tile_w, tile_h, ptr(alphas), ptr(last_ids),
                                 ptr(v_render),
                                 ptr(v_alphas), ptr(v_rec), ptr(v_abs), ptr(ctx.tile_order),
                                 1 if (ctx.geometry_only and not absgrad) else 0, stream_ptr(dev)), "gsx_raster_bwd")
        if absgrad and ctx.means2d_ref is not None:
            ctx.means2d_ref.absgrad = v_abs  # same side channel as gsplat (absgrad is off in gslam, rasterization.py:63)
        if getattr(ctx.means2d_ref, "_gsx_share_grad", False):
            # what means2d.retain_grad() would give (gslam/backend.py:326), as a view of v_rec instead of a copy kernel
            ctx.means2d_ref.grad = v_rec[..., 0:2]
        v_bg = None
        if bg is not None and ctx.needs_input_grad[3]:
            v_bg = (v_render * (1.0 - alphas)).sum(dim=(1, 2))
        return (v_rec, v_rec[..., 0:2], v_rec[..., 2:5], v_bg) + (None,) * 12


class _PackRecords(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities):
        means2d, conics = _f32c(means2d, "means2d"), _f32c(conics, "conics")
        colors, opacities = _f32c(colors, "colors"), _f32c(opacities, "opacities")
        Cn, N, ch = colors.shape
        rs = record_stride(ch)
        rec = torch.empty(Cn, N, rs, dtype=torch.float32, device=colors.device)
        check(lib.gsx_pack_records(ptr(means2d), ptr(conics), ptr(opacities), ptr(colors), N, Cn, ch, ptr(rec),
                                   stream_ptr(colors.device)), "gsx_pack_records")
        ctx.ch = ch
        return rec

    @staticmethod
    def backward(ctx, v_rec):
        if v_rec is None:
            return None, None, None, None
        ch = ctx.ch
        # xy / conic columns are delivered through the pass-through inputs of _RasterizeRecords
        return None, None, v_rec[..., 6:6 + ch], v_rec[..., 5]


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, image_width: int,
                        image_height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
                        backgrounds: Optional[Tensor] = None, masks: Optional[Tensor] = None, packed: bool = False,
                        absgrad: bool = False, visibility_min_T: float = 0.5, offsets_have_end: bool = False,
                        tile_order: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """The gsplat FORK's op as called at gslam/rasterization.py:325-339: returns the 3-tuple
    (render_colors[C,H,W,CH], render_alphas[C,H,W,1], n_touched[C,N] int32).
    offsets_have_end (extension, sync-free callers): ``isect_offsets`` is the flat int32 [T+1] array of isect_bin_sort and
    ``flatten_ids`` a capacity-sized buffer - tile ranges are clamped to it, nothing is read back."""
    if packed:
        # packed=True (gslam/rasterization.py:336): [nnz, ...] arrays indexed by flatten_ids.  The kernels index rows by
        # flatten id only (the camera comes from the tile), so the packed rows, padded to a multiple of C, ARE a [C, N']
        # layout with the same row numbers; n_touched comes back per packed row.
        Cn = int(isect_offsets.shape[0]) if not offsets_have_end else None
        if Cn is None:
            raise NotImplementedError("packed rasterize_to_pixels with flat offsets")
        nnz = int(means2d.shape[0])
        Np = max(1, -(-nnz // Cn))
        pad = Cn * Np - nnz

        def as_dense(t):
            if pad:
                t = torch.cat([t, torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)], 0)
            return t.reshape((Cn, Np) + tuple(t.shape[1:]))
        render, alphas, nt = rasterize_to_pixels(as_dense(means2d), as_dense(conics), as_dense(colors), as_dense(opacities),
                                                 image_width, image_height, tile_size, isect_offsets, flatten_ids,
                                                 backgrounds=backgrounds, masks=masks, packed=False, absgrad=absgrad,
                                                 visibility_min_T=visibility_min_T, tile_order=tile_order)
        return render, alphas, nt.reshape(-1)[:nnz]
    if masks is not None:
        raise NotImplementedError("tile masks are never passed by the reference")
    if tile_size != TILE:
        raise NotImplementedError("tile_size must be 16")
    ch_total = colors.shape[-1]
    renders, alphas, n_touched = [], None, None
    for c0 in range(0, ch_total, 5):
        cols = colors[..., c0:c0 + 5]
        bg = None if backgrounds is None else backgrounds[..., c0:c0 + 5]
        ch = cols.shape[-1]
        rec = _PackRecords.apply(means2d, conics, cols, opacities)
        r, a, nt, _ = _RasterizeRecords.apply(rec, means2d, conics, bg, isect_offsets, flatten_ids, ch,
                                              int(image_width), int(image_height), float(visibility_min_T),
                                              bool(absgrad), bool(offsets_have_end), True, None, tile_order, False)
        renders.append(r)
        if alphas is None:
            alphas, n_touched = a, nt
    render = renders[0] if len(renders) == 1 else torch.cat(renders, dim=-1)
    return render, alphas, n_touched


# ---------------------------------------------------------------------------------------------------------------
# K13
# ---------------------------------------------------------------------------------------------------------------
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs, radii):
        dirs, coeffs = _f32c(dirs, "dirs"), _f32c(coeffs, "coeffs")
        Cn, N = dirs.shape[:2]
        Kc = coeffs.shape[1]
        radii = None if radii is None else radii.contiguous()
        colors = torch.empty(Cn, N, 3, dtype=torch.float32, device=dirs.device)
        check(lib.gsx_sh_fwd(degree, ptr(dirs), ptr(coeffs), ptr(radii), N, Cn, Kc, ptr(colors),
                             stream_ptr(dirs.device)), "gsx_sh_fwd")
        ctx.save_for_backward(dirs, coeffs, radii)
        ctx.degree = degree
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        if v_colors is None:
            return None, None, None, None
        dirs, coeffs, radii = ctx.saved_tensors
        Cn, N = dirs.shape[:2]
        Kc = coeffs.shape[1]
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if ctx.needs_input_grad[1] else None
        check(lib.gsx_sh_bwd(ctx.degree, ptr(dirs), ptr(coeffs), ptr(radii), ptr(v_colors.contiguous()), N, Cn, Kc,
                             ptr(v_coeffs), ptr(v_dirs), stream_ptr(dirs.device)), "gsx_sh_bwd")
        return None, v_dirs, v_coeffs, None


class _SphericalHarmonicsFromMeans(torch.autograd.Function):
    """colours of (camera, Gaussian) pairs from the means and the camera centres: the view directions never exist as an array"""

    @staticmethod
    def forward(ctx, degree, means, campos, coeffs, radii):
        means, campos, coeffs = _f32c(means, "means"), _f32c(campos, "campos"), _f32c(coeffs, "coeffs")
        N, Cn, Kc = means.shape[0], campos.shape[0], coeffs.shape[1]
        radii = None if radii is None else radii.contiguous()
        colors = torch.empty(Cn, N, 3, dtype=torch.float32, device=means.device)
        check(lib.gsx_sh_fwd_means(degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii), N, Cn, Kc, ptr(colors),
                                   stream_ptr(means.device)), "gsx_sh_fwd_means")
        ctx.save_for_backward(means, campos, coeffs, radii)
        ctx.degree = degree
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        if v_colors is None:
            return None, None, None, None, None
        means, campos, coeffs, radii = ctx.saved_tensors
        N, Cn, Kc = means.shape[0], campos.shape[0], coeffs.shape[1]
        v_coeffs = torch.empty_like(coeffs)
        v_means = torch.empty_like(means) if ctx.needs_input_grad[1] else None
        v_campos = torch.zeros_like(campos) if ctx.needs_input_grad[2] else None
        check(lib.gsx_sh_bwd_means(ctx.degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii), ptr(v_colors.contiguous()),
                                   N, Cn, Kc, ptr(v_coeffs), ptr(v_means), ptr(v_campos), stream_ptr(means.device)),
              "gsx_sh_bwd_means")
        return None, v_means, v_campos, v_coeffs, None


def spherical_harmonics_from_means(degrees_to_use: int, means: Tensor, campos: Tensor, coeffs: Tensor,
                                   masks: Optional[Tensor] = None) -> Tensor:
    """spherical_harmonics(degree, means[None] - campos[:, None], coeffs, masks) without the [C,N,3] direction array:
    means [N,3], campos [C,3] (camera centres), coeffs [N,K,3] -> [C,N,3]; differentiable in all three"""
    radii = None if masks is None else masks.to(torch.int32)
    return _SphericalHarmonicsFromMeans.apply(int(degrees_to_use), means, campos, coeffs, radii)


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """gsplat spherical_harmonics: dirs [C,N,3] (un-normalised), coeffs [N,K,3] -> [C,N,3] = max(0, SH + 0.5).
    ``masks`` may be a bool [C,N] or the int32 radii (evaluated where > 0)."""
    radii = None
    if masks is not None:
        radii = masks.to(torch.int32)
    return _SphericalHarmonics.apply(int(degrees_to_use), dirs, coeffs, radii)


@torch.no_grad()
def rebuild_isect_ids(out, M: int) -> Tensor:
    """Sorted gsplat keys (cam << (32+tile_n_bits) | tile << 32 | float_bits(depth)) for a sync-free render, built on
    demand from flatten_ids / depths / offsets (nobody on the gslam hot path reads them)."""
    flat = out._flatten_ids.long()
    dev = flat.device
    n_tiles = out.tile_width * out.tile_height
    tnb = int(n_tiles).bit_length()
    dbits = out.depths.detach().reshape(-1).view(torch.int32)[flat].long() & 0xFFFFFFFF
    offsets = out.isect_offsets.reshape(-1).long()
    tile = torch.searchsorted(offsets, torch.arange(M, device=dev), right=True) - 1
    cam, tl = tile // n_tiles, tile % n_tiles
    return (cam << (32 + tnb)) | (tl << 32) | dbits
