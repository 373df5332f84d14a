"""ctypes binding of gslam_amd/libgsx.so (C ABI: include/gsx.h).

The library is the product: if it is missing, fails to load or lacks a symbol, importing this module raises - there
is no CPU / PyTorch fallback behind these ops.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads torch's bundled libamdhip64 first so libgsx binds to the same HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsx.so")

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> (restype, argtypes); mirrors include/gsx.h one to one
PROTOTYPES = {
    "gsx_version": (i32, []),
    "gsx_last_error": (C.c_char_p, []),
    "gsx_record_stride": (i32, [i32]),
    "gsx_read_i64": (i32, [vp, C.POINTER(i64), vp]),
    "gsx_project_fwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp,
                              i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "gsx_project_fwd_rects": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp,
                              i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gsx_project_bwd_workspace_bytes": (i64, [i64, i64]),
    "gsx_project_bwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, i32, vp, vp, i64, vp, vp, i64, vp,
                              vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "gsx_project_bwd_range": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, i32, vp, vp, i64, vp, vp, i64, vp,
                                    vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "gsx_quat_scale_to_covar_preci": (i32, [vp, vp, i64, vp, vp, vp]),
    "gsx_pack_records": (i32, [vp, vp, vp, vp, i64, i64, i32, vp, vp]),
    "gsx_isect_count": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "gsx_scan_workspace_bytes": (i64, [i64]),
    "gsx_isect_scan": (i32, [vp, i64, vp, vp, i64, vp]),
    "gsx_isect_emit": (i32, [vp, vp, vp, vp, i64, i64, i32, i32, i64, vp, vp, vp]),
    "gsx_isect_offset_encode": (i32, [vp, i64, i64, i32, i32, vp, vp]),
    "gsx_isect_bin_workspace_bytes": (i64, [i64, i32, i32, i64]),
    "gsx_isect_bin_workspace_bytes_n": (i64, [i64, i64, i32, i32, i64]),
    "gsx_isect_bin_sort": (i32, [vp, vp, vp, i64, i64, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "gsx_isect_bin_sort_rects": (i32, [vp, vp, i64, i64, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "gsx_raster_fwd": (i32, [vp, i32, vp, vp, vp, i64, i32, i64, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]),
    "gsx_raster_bwd": (i32, [vp, i32, vp, vp, vp, i64, i32, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "gsx_sh_fwd": (i32, [i32, vp, vp, vp, i64, i64, i32, vp, vp]),
    "gsx_sh_bwd": (i32, [i32, vp, vp, vp, vp, i64, i64, i32, vp, vp, vp]),
    "gsx_sh_fwd_means": (i32, [i32, vp, vp, vp, vp, i64, i64, i32, vp, vp]),
    "gsx_sh_bwd_means": (i32, [i32, vp, vp, vp, vp, vp, i64, i64, i32, vp, vp, vp, vp]),
    "gsx_ssim_workspace_bytes": (i64, [i64, i32, i32, i32]),
    "gsx_ssim_fwd": (i32, [vp, vp, i64, i32, i32, i32, C.POINTER(i64), C.POINTER(i64), i32, vp, vp, vp, vp, vp, i64, vp]),
    "gsx_ssim_bwd": (i32, [vp, vp, i64, i32, i32, i32, C.POINTER(i64), C.POINTER(i64), i32, vp, vp, vp, vp, f32, vp, vp]),
    "gsx_map_loss_workspace_bytes": (i64, [i64, i32, i32]),
    "gsx_map_loss": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, i64, vp]),
    "gsx_isotropic_workspace_bytes": (i64, [i64]),
    "gsx_isotropic_loss": (i32, [vp, vp, i64, f32, vp, vp, vp, i64, vp]),
    "gsx_isotropic_loss_acc": (i32, [vp, vp, i64, f32, vp, vp, vp, i64, vp]),
    "gsx_ssim_partials": (i64, [i64, i32, i32, i32]),
    "gsx_loss_finish": (i32, [vp, i64, i32, i32, vp, i64, vp, i64, C.POINTER(f32), C.POINTER(f32), f32, f32, vp, vp, vp,
                              vp]),
    "gsx_combine_terms": (i32, [i32, C.POINTER(vp), C.POINTER(f32), C.POINTER(f32), f32, f32, vp, vp]),
    "gsx_opacity_decay": (i32, [vp, vp, i64, i32, f32, vp]),
    "gsx_warp_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    "gsx_warp_bwd_workspace_bytes": (i64, [i32, i32]),
    "gsx_warp_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp]),
    "gsx_pose_zhou_fwd": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), vp, vp]),
    "gsx_pose_zhou_bwd": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), vp, C.POINTER(vp),
                                C.POINTER(vp), vp]),
    "gsx_pose_zhou_bwd_partials": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), vp, i64, vp,
                                         C.POINTER(vp), C.POINTER(vp), vp]),
    "gsx_adam_multi": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64),
                             C.POINTER(f32), f32, f32, f32, i64, vp, vp]),
    "gsx_adam_multi_steps": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64),
                                   C.POINTER(f32), f32, f32, f32, C.POINTER(vp), vp]),
    "gsx_adam_multi_steps_decay": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                         C.POINTER(i64), C.POINTER(f32), f32, f32, f32, C.POINTER(vp), i32, vp, i32, f32,
                                         vp]),
    "gsx_counters_add": (i32, [i32, C.POINTER(vp), i64, vp]),
    "gsx_adam_multi_steps_gated": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                         C.POINTER(i64), C.POINTER(f32), f32, f32, f32, C.POINTER(vp), i32, vp, i32, f32,
                                         vp, vp]),
    "gsx_counters_add_gated": (i32, [i32, C.POINTER(vp), i64, vp, vp]),
    "gsx_status_flag": (i32, [vp, i32, i32, vp, vp]),
    "gsx_front_pose_bwd_tail_words": (i64, []),
    "gsx_front_pose_bwd_tail": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, f32, f32, f32, i32, vp, i64, vp, i64, vp, vp, vp, vp, vp,
                                      vp, vp, vp, i64, f32, vp, vp]),
    "gsx_ssim_bwd_map_loss_rows": (i64, [i64, i32, i32]),
    "gsx_ssim_bwd_map_loss": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, f32, f32, i32, vp, vp, vp, vp, f32, vp,
                                    vp, i64, vp]),
    "gsx_range_copy": (i32, [i32, C.POINTER(vp), C.POINTER(i64), i32, vp, i32, vp]),
    "gsx_track_opt_state_bytes": (i64, []),
    "gsx_track_opt_init": (i32, [vp, i32, i32, f32, C.c_double, i32, i32, i32, C.c_double, C.c_double, vp]),
    "gsx_track_opt_advance": (i32, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), vp, vp]),
    "gsx_track_opt_report": (i32, [vp, vp, vp]),
    "gsx_track_opt_tail": (i32, [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, vp]),
    "gsx_project_bwd_blocks": (i64, [i64]),
    "gsx_window_opt_state_bytes": (i64, []),
    "gsx_window_opt_init": (i32, [vp, i32, i32, f32, C.c_double, i32, i32, i32, C.c_double, C.c_double, vp]),
    "gsx_window_opt_advance": (i32, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), vp, vp]),
    "gsx_window_opt_report": (i32, [vp, vp, vp]),
    "gsx_window_opt_tail": (i32, [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, vp]),
    "gsx_gather_rows": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), vp, i64, i64, vp]),
    "gsx_concat_rows": (i32, [i32, C.POINTER(vp), i64, C.POINTER(vp), i64, C.POINTER(vp), C.POINTER(i32), vp]),
    "gsx_selftest": (i32, [vp, i64, vp]),
    "gsx_stream_create": (i32, [C.POINTER(vp)]),
    "gsx_stream_create_masked": (i32, [C.POINTER(vp), vp, i32]),
    "gsx_stream_destroy": (i32, [vp]),
    "gsx_stream_synchronize": (i32, [vp]),
    "gsx_stream_wait_stream": (i32, [vp, vp]),
    "gsx_graph_begin": (i32, [vp, i32]),
    "gsx_graph_end": (i32, [vp, C.POINTER(vp), C.POINTER(i64)]),
    "gsx_graph_abort": (i32, [vp]),
    "gsx_graph_launch": (i32, [vp, vp]),
    "gsx_graph_launch_n": (i32, [vp, i32, vp]),
    "gsx_graph_destroy": (i32, [vp]),
    "gsx_host_alloc": (i32, [C.POINTER(vp), C.POINTER(vp), i64]),
    "gsx_host_free": (i32, [vp]),
    "gsx_zero_words": (i32, [vp, i64, vp]),
    "gsx_probe_wg_placement": (i32, [i32, i32, i32, vp, vp]),
    "gsx_raster_fwd_track_loss": (i32, [vp, vp, vp, vp, i64, i32, i64, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gsx_tile_balance": (i32, [vp, i64, f32, f32, i32, vp, vp]),
    "gsx_raster_track_fused": (i32, [vp, vp, vp, vp, i64, i32, i64, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gsx_raster_track_fused_sorting": (i32, [vp, vp, vp, vp, i64, i32, i64, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp,
                                             vp, vp, C.c_uint32, vp, f32, vp, vp, vp]),
    "gsx_raster_track_fused_near": (i32, [vp, vp, vp, vp, i64, i32, i64, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp,
                                          vp, vp, C.c_uint32, vp, f32, vp, vp, vp, vp, vp, i64, i64, i32, vp]),
    "gsx_front_fwd_near": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp,
                                 vp, vp, f32, f32, i32, vp, i64, vp, vp, vp]),
    "gsx_raster_track_fused_lds_bytes": (i64, []),
    "gsx_front_rows_layout": (i32, [i64, i64, i32, i32, i64, C.POINTER(i64)]),
    "gsx_raster_track_fused_rows": (i32, [vp, vp, vp, i64, i64, i64, i32, i32, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32,
                                          vp, vp, vp, vp, vp, vp, i64, vp]),
    "gsx_front_keys": (i32, [i64, i64, i32, i32, i64, i32, C.POINTER(i64)]),
    "gsx_front_workspace_bytes": (i64, [i64, i64, i32, i32, i64]),
    "gsx_front_rows": (i64, [i64, i64, i32, i32]),
    "gsx_front_workspace_bytes_cand": (i64, [i64, i64, i32, i32, i64]),
    "gsx_front_candidates": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, i32, vp, vp, vp, f32, f32, i64,
                                   vp, i64, vp]),
    "gsx_front_cand_layout": (i32, [i64, i64, i32, i32, i64, C.POINTER(i64)]),
    "gsx_front_layout": (i32, [i64, i64, i32, i32, i64, C.POINTER(i64)]),
    "gsx_front_pose_bwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, i32, vp, i64, vp, i64, vp, vp]),
    "gsx_front_fwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                            vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, vp, i64, vp]),
}


class GsxError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m gslam_amd.csrc.build` (hipcc, gfx950). "
            "gslam_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise GsxError(f"{what} failed (rc={rc}): {lib.gsx_last_error().decode(errors='replace')}")


def ptr(t: torch.Tensor | None):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(*ts: torch.Tensor | None):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise GsxError("gslam_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback")
        if not t.is_contiguous():
            raise GsxError("gslam_amd C-ABI expects contiguous tensors")
