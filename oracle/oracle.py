"""ctypes/numpy front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
gslam_amd/ does.  See oracle/gsx_oracle.c for what is restated and why parity is "unpinned" for the external
CUDA kernels.

Every function takes and returns numpy arrays.  ``Oracle(np.float32)`` is the oracle proper; ``Oracle(np.float64)``
is the same code compiled with REAL=double, used for finite-difference checks of the hand-derived VJPs.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force: bool = False) -> None:
    """Compile oracle/gsx_oracle.c (gcc) into oracle/_build/ if missing or stale."""
    src = os.path.join(_HERE, "gsx_oracle.c")
    outs = [os.path.join(_BUILD, f"libgsx_oracle_{s}.so") for s in ("f32", "f64", "f32_omp")]
    stale = force or any((not os.path.exists(o)) or os.path.getmtime(o) < os.path.getmtime(src) for o in outs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "all"], check=True, capture_output=True)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, dtype=np.float32, threads: bool = False):
        """threads=True: the OpenMP build (all host cores) - for bench.py's cpu_baseline leg only; its gradient sums are
        atomics whose order depends on the schedule, so the tests use the serial builds"""
        build()
        self.dtype = np.dtype(dtype)
        suffix = "f64" if self.dtype == np.float64 else ("f32_omp" if threads else "f32")
        self.lib = C.CDLL(os.path.join(_BUILD, f"libgsx_oracle_{suffix}.so"))
        self.real = C.c_double if self.dtype == np.float64 else C.c_float
        assert self.lib.gsxo_sizeof_real() == self.dtype.itemsize
        self.threads = int(self.lib.gsxo_omp_threads())

    def set_threads(self, n: int) -> int:
        """OpenMP build only: threads the next calls use (-> the number in effect)"""
        self.lib.gsxo_omp_set_threads(int(n))
        self.threads = int(self.lib.gsxo_omp_threads())
        return self.threads

    # -- helpers -----------------------------------------------------------
    def r(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None:
            assert a.shape == tuple(shape), (a.shape, shape)
        return a

    @staticmethod
    def i32(a):
        return np.ascontiguousarray(a, dtype=np.int32)

    def _call(self, name, *args):
        fn = getattr(self.lib, name)
        fn.restype = C.c_int
        rc = fn(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed rc={rc}")

    # -- K10 ----------------------------------------------------------------
    def quat_scale_to_covar_preci(self, quats, scales, compute_preci=True):
        quats, scales = self.r(quats), self.r(scales)
        n = quats.shape[0]
        cov = np.empty((n, 3, 3), self.dtype)
        pre = np.empty((n, 3, 3), self.dtype) if compute_preci else None
        self._call("gsxo_quat_scale_to_covar_preci", C.c_int64(n), _p(quats), _p(scales), _p(cov), _p(pre))
        return cov, pre

    # -- K1 / K2 --------------------------------------------------------------
    def project_fwd(self, means, quats, scales, viewmats, Ks, W, H, eps2d=0.3, near=0.01, far=1e10,
                    radius_clip=0.0, calc_compensations=False):
        means, quats, scales = self.r(means), self.r(quats), self.r(scales)
        viewmats, Ks = self.r(viewmats), self.r(Ks)
        N, Cn = means.shape[0], viewmats.shape[0]
        radii = np.empty((Cn, N), np.int32)
        means2d = np.empty((Cn, N, 2), self.dtype)
        depths = np.empty((Cn, N), self.dtype)
        conics = np.empty((Cn, N, 3), self.dtype)
        comps = np.empty((Cn, N), self.dtype) if calc_compensations else None
        self._call("gsxo_project_fwd", C.c_int64(N), C.c_int64(Cn), _p(means), _p(quats), _p(scales), _p(viewmats),
                   _p(Ks), C.c_int(W), C.c_int(H), self.real(eps2d), self.real(near), self.real(far),
                   self.real(radius_clip), _p(radii), _p(means2d), _p(depths), _p(conics), _p(comps))
        return radii, means2d, depths, conics, comps

    def project_bwd(self, means, quats, scales, viewmats, Ks, W, H, radii, v_means2d, v_depths, v_conics,
                    v_comps=None, eps2d=0.3, near=0.01, far=1e10, viewmats_grad=True):
        means, quats, scales = self.r(means), self.r(quats), self.r(scales)
        viewmats, Ks = self.r(viewmats), self.r(Ks)
        N, Cn = means.shape[0], viewmats.shape[0]
        v_means2d, v_conics = self.r(v_means2d, (Cn, N, 2)), self.r(v_conics, (Cn, N, 3))
        v_depths = None if v_depths is None else self.r(v_depths, (Cn, N))
        v_comps = None if v_comps is None else self.r(v_comps, (Cn, N))
        v_means = np.zeros((N, 3), self.dtype)
        v_quats = np.zeros((N, 4), self.dtype)
        v_scales = np.zeros((N, 3), self.dtype)
        v_view = np.zeros((Cn, 4, 4), self.dtype) if viewmats_grad else None
        self._call("gsxo_project_bwd", C.c_int64(N), C.c_int64(Cn), _p(means), _p(quats), _p(scales), _p(viewmats),
                   _p(Ks), C.c_int(W), C.c_int(H), self.real(eps2d), self.real(near), self.real(far),
                   _p(self.i32(radii)), _p(v_means2d), _p(v_depths), _p(v_conics), _p(v_comps), _p(v_means),
                   _p(v_quats), _p(v_scales), _p(v_view))
        return v_means, v_quats, v_scales, v_view

    # -- K3..K7 ---------------------------------------------------------------
    def isect_tiles(self, means2d, radii, depths, tile_size, tile_w, tile_h, sort=True):
        means2d = np.ascontiguousarray(means2d, np.float32)
        depths = np.ascontiguousarray(depths, np.float32)
        radii = self.i32(radii)
        Cn, N = radii.shape
        tpg = np.empty((Cn, N), np.int32)
        cum = np.empty((Cn * N,), np.int64)
        self._call("gsxo_isect_count", C.c_int64(Cn), C.c_int64(N), _p(means2d), _p(radii), C.c_int(tile_size),
                   C.c_int(tile_w), C.c_int(tile_h), _p(tpg), _p(cum))
        M = int(cum[-1]) if cum.size else 0
        isect_ids = np.empty((M,), np.int64)
        flatten_ids = np.empty((M,), np.int32)
        self._call("gsxo_isect_emit_sort", C.c_int64(Cn), C.c_int64(N), _p(means2d), _p(radii), _p(depths), _p(cum),
                   C.c_int(tile_size), C.c_int(tile_w), C.c_int(tile_h), C.c_int(1 if sort else 0), _p(isect_ids),
                   _p(flatten_ids))
        return tpg, isect_ids, flatten_ids

    def isect_offset_encode(self, isect_ids, Cn, tile_w, tile_h):
        isect_ids = np.ascontiguousarray(isect_ids, np.int64)
        off = np.empty((Cn, tile_h, tile_w), np.int32)
        self._call("gsxo_isect_offset_encode", C.c_int64(isect_ids.shape[0]), _p(isect_ids), C.c_int64(Cn),
                   C.c_int(tile_w), C.c_int(tile_h), _p(off))
        return off

    # -- K8 / K9 --------------------------------------------------------------
    def raster_fwd(self, means2d, conics, colors, opacities, backgrounds, W, H, tile_size, offsets, flatten_ids,
                   vis_min_T=0.5):
        means2d, conics, colors, opacities = self.r(means2d), self.r(conics), self.r(colors), self.r(opacities)
        Cn, N, CH = colors.shape
        backgrounds = None if backgrounds is None else self.r(backgrounds, (Cn, CH))
        offsets, flatten_ids = self.i32(offsets), self.i32(flatten_ids)
        tile_h, tile_w = offsets.shape[1:]
        render = np.empty((Cn, H, W, CH), self.dtype)
        alphas = np.empty((Cn, H, W, 1), self.dtype)
        last_ids = np.empty((Cn, H, W), np.int32)
        n_touched = np.empty((Cn, N), np.int32)
        self._call("gsxo_raster_fwd", C.c_int64(Cn), C.c_int64(N), C.c_int(CH), _p(means2d), _p(conics), _p(colors),
                   _p(opacities), _p(backgrounds), C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tile_w),
                   C.c_int(tile_h), _p(offsets), _p(flatten_ids), C.c_int64(flatten_ids.shape[0]),
                   self.real(vis_min_T), _p(render), _p(alphas), _p(last_ids), _p(n_touched))
        return render, alphas, last_ids, n_touched

    def raster_bwd(self, means2d, conics, colors, opacities, backgrounds, W, H, tile_size, offsets, flatten_ids,
                   alphas, last_ids, v_render, v_alphas, absgrad=False):
        means2d, conics, colors, opacities = self.r(means2d), self.r(conics), self.r(colors), self.r(opacities)
        Cn, N, CH = colors.shape
        backgrounds = None if backgrounds is None else self.r(backgrounds, (Cn, CH))
        offsets, flatten_ids, last_ids = self.i32(offsets), self.i32(flatten_ids), self.i32(last_ids)
        tile_h, tile_w = offsets.shape[1:]
        alphas = self.r(alphas).reshape(Cn, H, W)
        v_render = self.r(v_render, (Cn, H, W, CH))
        v_alphas = self.r(v_alphas).reshape(Cn, H, W)
        v_means2d = np.zeros((Cn, N, 2), self.dtype)
        v_conics = np.zeros((Cn, N, 3), self.dtype)
        v_colors = np.zeros((Cn, N, CH), self.dtype)
        v_opac = np.zeros((Cn, N), self.dtype)
        v_abs = np.zeros((Cn, N, 2), self.dtype) if absgrad else None
        self._call("gsxo_raster_bwd", C.c_int64(Cn), C.c_int64(N), C.c_int(CH), _p(means2d), _p(conics), _p(colors),
                   _p(opacities), _p(backgrounds), C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tile_w),
                   C.c_int(tile_h), _p(offsets), _p(flatten_ids), C.c_int64(flatten_ids.shape[0]), _p(alphas),
                   _p(last_ids), _p(v_render), _p(v_alphas), _p(v_means2d), _p(v_conics), _p(v_colors), _p(v_opac),
                   _p(v_abs))
        return v_means2d, v_conics, v_colors, v_opac, v_abs

    # -- K13 ------------------------------------------------------------------
    def sh_fwd(self, deg, dirs, coeffs, radii=None):
        dirs, coeffs = self.r(dirs), self.r(coeffs)
        Cn, N = dirs.shape[:2]
        Kc = coeffs.shape[1]
        radii = None if radii is None else self.i32(radii)
        colors = np.empty((Cn, N, 3), self.dtype)
        self._call("gsxo_sh_fwd", C.c_int64(Cn), C.c_int64(N), C.c_int(Kc), C.c_int(deg), _p(dirs), _p(coeffs),
                   _p(radii), _p(colors))
        return colors

    def sh_bwd(self, deg, dirs, coeffs, v_colors, radii=None):
        dirs, coeffs, v_colors = self.r(dirs), self.r(coeffs), self.r(v_colors)
        Cn, N = dirs.shape[:2]
        Kc = coeffs.shape[1]
        radii = None if radii is None else self.i32(radii)
        v_coeffs = np.zeros_like(coeffs)
        v_dirs = np.empty_like(dirs)
        self._call("gsxo_sh_bwd", C.c_int64(Cn), C.c_int64(N), C.c_int(Kc), C.c_int(deg), _p(dirs), _p(coeffs),
                   _p(radii), _p(v_colors), _p(v_coeffs), _p(v_dirs))
        return v_coeffs, v_dirs

    # -- K11 / K12 --------------------------------------------------------------
    def ssim_maps(self, img1, img2, need_grad=True):
        img1, img2 = self.r(img1), self.r(img2)
        B, CH, H, W = img1.shape
        m = np.empty_like(img1)
        d1 = np.empty_like(img1) if need_grad else None
        d2 = np.empty_like(img1) if need_grad else None
        d3 = np.empty_like(img1) if need_grad else None
        self._call("gsxo_ssim_fwd", C.c_int64(B), C.c_int(CH), C.c_int(H), C.c_int(W), _p(img1), _p(img2), _p(m),
                   _p(d1), _p(d2), _p(d3))
        return m, d1, d2, d3

    def fused_ssim(self, img1, img2, padding="same"):
        """-> (scalar mean, dL/dimg1 for dL/dout = 1)"""
        img1, img2 = self.r(img1), self.r(img2)
        m, d1, d2, d3 = self.ssim_maps(img1, img2)
        B, CH, H, W = img1.shape
        dmap = np.zeros_like(m)
        if padding == "valid":
            mv = m[:, :, 5:-5, 5:-5]
            dmap[:, :, 5:-5, 5:-5] = 1.0 / mv.size
        else:
            mv = m
            dmap[...] = 1.0 / mv.size
        val = mv.mean(dtype=np.float64)
        g = np.empty_like(img1)
        self._call("gsxo_ssim_bwd", C.c_int64(B), C.c_int(CH), C.c_int(H), C.c_int(W), _p(img1), _p(img2), _p(dmap),
                   _p(d1), _p(d2), _p(d3), _p(g))
        return self.dtype.type(val), g

    # -- Warp ----------------------------------------------------------------------
    def warp_fwd(self, T, K, Kinv, c1, d1):
        T, K, Kinv, c1, d1 = self.r(T, (4, 4)), self.r(K, (3, 3)), self.r(Kinv, (3, 3)), self.r(c1), self.r(d1)
        H, W = d1.shape
        res = np.empty((H, W, 3), self.dtype)
        nw = np.empty((H, W, 2), self.dtype)
        mask = np.empty((H, W), np.uint8)
        self._call("gsxo_warp_fwd", C.c_int(H), C.c_int(W), _p(T), _p(K), _p(Kinv), _p(c1), _p(d1), _p(res), _p(nw),
                   _p(mask))
        return res, nw[None], mask.astype(bool)

    def warp_bwd(self, T, K, Kinv, c1, d1, v_result, v_nwarps=None):
        T, K, Kinv, c1, d1 = self.r(T, (4, 4)), self.r(K, (3, 3)), self.r(Kinv, (3, 3)), self.r(c1), self.r(d1)
        H, W = d1.shape
        v_result = self.r(v_result, (H, W, 3))
        v_nwarps = None if v_nwarps is None else self.r(v_nwarps).reshape(H, W, 2)
        vT = np.empty((4, 4), self.dtype)
        self._call("gsxo_warp_bwd", C.c_int(H), C.c_int(W), _p(T), _p(K), _p(Kinv), _p(c1), _p(d1), _p(v_result),
                   _p(v_nwarps), _p(vT))
        return vT

    # -- Adam ----------------------------------------------------------------------------
    def adam(self, p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
        """in-place on copies; returns (p, m, v)"""
        p, g, m, v = self.r(p).copy(), self.r(g), self.r(m).copy(), self.r(v).copy()
        self._call("gsxo_adam", C.c_int64(p.size), _p(p), _p(g), _p(m), _p(v), self.real(lr), self.real(beta1),
                   self.real(beta2), self.real(eps), C.c_int64(step))
        return p, m, v

    # -- gslam/rasterization.py host logic, restated on top of the ops above ---------------
    def gslam_rasterization(self, means, quats, log_scales, logit_opacities, logit_colors, viewmats, Ks, W, H,
                            render_mode="RGB", log_uncertainties=None, backgrounds=None, eps2d=0.3, near=0.01,
                            far=1e10, radius_clip=0.0, tile_size=16, vis_min_T=0.5, scales_override=None):
        """Restates gslam/rasterization.py:121-360 for packed=False, classic mode.

        Returns a dict with the RasterizationOutput fields.  ``scales_override`` lets a caller inject
        post-activation scales computed elsewhere (e.g. torch.exp on the GPU) so that integer outputs can be
        compared bit-for-bit independent of libm's expf.
        """
        dt = self.dtype
        sig = lambda x: (1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))).astype(dt)
        opac = sig(logit_opacities)                                   # rasterization.py:145
        colors = sig(logit_colors)                                    # :146
        scales = np.exp(np.asarray(log_scales, dt)) if scales_override is None else np.asarray(scales_override, dt)
        betas = None
        if log_uncertainties is not None:
            betas = np.maximum(np.exp(np.asarray(log_uncertainties, dt)), dt.type(0.01))   # :149
        radii, means2d, depths, conics, _ = self.project_fwd(means, quats, scales, viewmats, Ks, W, H, eps2d, near,
                                                             far, radius_clip)
        Cn, N = radii.shape
        opac_c = np.broadcast_to(opac[None], (Cn, N)).copy()          # :187
        cols = np.broadcast_to(colors[None], (Cn, N, colors.shape[-1])).copy()   # :225
        bg = None if backgrounds is None else np.asarray(backgrounds, dt)
        depth_index = betas_index = None
        if render_mode in ("RGB+D", "RGB+ED"):
            cols = np.concatenate([cols, depths[..., None]], -1)     # :235
            if bg is not None:
                bg = np.concatenate([bg, np.zeros((Cn, 1), dt)], -1)
            depth_index = cols.shape[-1] - 1
        elif render_mode in ("D", "ED"):
            cols = depths[..., None].copy()
            if bg is not None:
                bg = np.zeros((Cn, 1), dt)
            depth_index = 0
        if betas is not None:
            cols = np.concatenate([cols, np.broadcast_to(betas[None], (Cn, N))[..., None]], -1)   # :250
            if bg is not None:
                bg = np.concatenate([bg, np.full((Cn, 1), math.e, dt)], -1)   # :253  exp(1.0)
            betas_index = cols.shape[-1] - 1
        tile_w, tile_h = math.ceil(W / tile_size), math.ceil(H / tile_size)
        tpg, isect_ids, flatten_ids = self.isect_tiles(means2d, radii, depths, tile_size, tile_w, tile_h)
        offsets = self.isect_offset_encode(isect_ids, Cn, tile_w, tile_h)
        render, alphas, last_ids, n_touched = self.raster_fwd(means2d, conics, cols, opac_c, bg, W, H, tile_size,
                                                              offsets, flatten_ids, vis_min_T)
        out = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opac_c,
                   tiles_per_gauss=tpg, isect_ids=isect_ids, flatten_ids=flatten_ids, isect_offsets=offsets,
                   tile_width=tile_w, tile_height=tile_h, width=W, height=H, tile_size=tile_size, n_cameras=Cn,
                   alphas=alphas, n_touched=n_touched.astype(np.int64), last_ids=last_ids, colors_packed=cols,
                   backgrounds_packed=bg)
        if depth_index is not None:
            out["depthmaps"] = render[..., depth_index]
            if render_mode in ("ED", "RGB+ED"):
                # :342-344 raises KeyError('depthaps') in the reference; this is the intent its comment documents
                # ("normalize the accumulated depth to get the expected depth"), which the HIP path implements
                out["depthmaps"] = out["depthmaps"] / np.maximum(alphas[..., 0], dt.type(1e-10))
        if betas_index is not None:
            out["betas"] = render[..., betas_index]
        out["rgbs"] = render[..., :3] if render_mode not in ("D", "ED") else None
        out["render"] = render
        return out
