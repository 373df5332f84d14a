"""Builds gslam_amd/libgsx.so (gfx950) from the HIP sources in this directory with hipcc.

In-tree build: the .so travels to the GPU box with the repo snapshot; nothing is JIT-compiled at import time.
``python -m gslam_amd.csrc.build [--force]``.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libgsx.so")
OBJ = os.path.join(HERE, "_obj")
ARCH = "gfx950"

# -ffp-contract=off: integer outputs (radii, tile rectangles, sort keys) must match the CPU oracle bit for bit.
SOURCES = {
    "project.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],     # (SLP: see raster.hip; BA iteration -1.1 % at 500 k x 8)
    "isect.hip": ["-ffp-contract=off"],
    "isect_bin.hip": ["-ffp-contract=off"],
    # -fno-slp-vectorize: LLVM's SLP pass pairs scalar fp32 operations of the rasteriser loops into v_pk_*_f32 and pays for every
    # pair with register moves (fused tracking kernel: 808 -> 761 VALU instructions, 147 -> 88 moves, 66 -> 62 VGPRs without it);
    # same-box A/B (tools/dbg/ab_flags.sh): tracking alone 150.3 -> 155.4 frames/s, headline 116.9 -> 120.1.  The packed colour
    # accumulation the kernels ask for explicitly (ext_vector types) is not affected.
    "raster.hip": ["-fno-slp-vectorize"],
    "ssim.hip": ["-fno-slp-vectorize"],                             # (-0.7 %)
    "loss.hip": ["-fno-slp-vectorize"],                             # (-0.5 %)
    "warp.hip": [],
    "pose.hip": [],
    "misc.hip": [],
    "track_opt.hip": [],
    "window_opt.hip": [],
    "maintain.hip": [],
    "runtime.hip": [],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-DNDEBUG"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, extra: list[str], force: bool) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(HERE, src), os.path.join(HERE, "gsx_common.h"), os.path.join(HERE, "raster_v4.inc"), os.path.join(HERE, "raster_fwd_state.inc"), os.path.join(HERE, "raster_fwd_chunks.inc"), os.path.join(HERE, "raster_fwd_finish.inc"), os.path.join(HERE, "tile_sort_lds.h"), os.path.join(HERE, "raster_bwd_loop.inc"), os.path.join(HERE, "raster_bwd_loop_tc.inc"), os.path.join(HERE, "track_opt.h"), os.path.join(HERE, "track_opt_impl.inc"), os.path.join(HERE, "track_tail.h"), os.path.join(HERE, "pose_chain.h"), os.path.join(HERE, "loss_pixel.h"), os.path.join(HERE, "tile_rect.h"), os.path.join(HERE, "pose_math.h"), os.path.join(HERE, "project_core.h"), os.path.join(HERE, "tile_balance.h"),
            os.path.join(PKG, "..", "include", "gsx.h"), os.path.abspath(__file__)]
    if force or _stale(obj, deps):
        cmd = ["hipcc", "-c", os.path.join(HERE, src), "-o", obj] + COMMON + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(lambda kv: _compile(kv[0], kv[1], force), SOURCES.items()))
    if force or _stale(OUT, objs):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
