"""Interleaved A/B timing of render stages in ONE process (cdna guide rule 24): v1 vs v2 raster kernels etc.
usage: python tools/ab_raster.py [N ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import ops  # noqa: E402
from gslam_amd.rasterization import rasterization  # noqa: E402
from gslam_amd.synthetic import make_cameras, make_scene  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100_000, 500_000]
    for N in sizes:
        W, H, C = 640, 480, 1
        sc = {k: v.to(dev) for k, v in make_scene(N, 0).items()}
        viewmats, Ks = make_cameras(C, W, H)
        viewmats, Ks = viewmats.to(dev), Ks.to(dev)
        for k in ("means", "quats", "scales", "opacities", "colors", "log_uncertainties"):
            sc[k].requires_grad_(True)
        out = rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats, Ks, W, H,
                            packed=False, render_mode="RGB+D", log_uncertainties=sc["log_uncertainties"],
                            backgrounds=torch.zeros(C, 3, device=dev))
        from gslam_amd.rasterization import validate
        if not validate(dev):      # capacity grown after a size jump: render again
            out = rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats, Ks,
                                W, H, packed=False, render_mode="RGB+D", log_uncertainties=sc["log_uncertainties"],
                                backgrounds=torch.zeros(C, 3, device=dev))
            assert validate(dev)
        M = out.flatten_ids.shape[0]
        print(f"N={N} M={M} visible={(out.radii > 0).sum().item()}")
        # direct stage calls
        from gslam_amd._lib import check, lib, ptr, stream_ptr
        rec = torch.empty(C, N, 12, device=dev)
        radii, m2d, dep, con, _, rec, tiles, _ = ops._Projection.apply(
            sc["means"].detach(), sc["quats"].detach(), sc["scales"].detach(), viewmats, Ks, sc["opacities"].detach(),
            sc["colors"].detach(), sc["log_uncertainties"].detach(), W, H, 0.3, 0.01, 1e10, 0.0, False, 1 | 2 | 4, True,
            True)
        off, flat = out.isect_offsets.contiguous(), out.flatten_ids.contiguous()
        bg = torch.zeros(C, 5, device=dev)
        bg[:, 4] = 2.718281828
        render = torch.empty(C, H, W, 5, device=dev)
        alphas = torch.empty(C, H, W, 1, device=dev)
        last = torch.empty(C, H, W, dtype=torch.int32, device=dev)
        nt = torch.zeros(C, N, dtype=torch.int32, device=dev)
        v_render = torch.randn(C, H, W, 5, device=dev)
        v_alpha = torch.randn(C, H, W, 1, device=dev)
        v_rec = torch.zeros(C, N, 12, device=dev)
        st = stream_ptr(dev)

        def fwd():
            check(lib.gsx_raster_fwd(ptr(rec), 5, ptr(bg), ptr(off), ptr(flat), M, 0, C, W, H, 40, 30, 0.5, ptr(render),
                                     ptr(alphas), ptr(last), ptr(nt), st), "fwd")

        def bwd():
            check(lib.gsx_raster_bwd(ptr(rec), 5, ptr(bg), ptr(off), ptr(flat), M, 0, C, W, H, 40, 30, ptr(alphas),
                                     ptr(last), ptr(v_render), ptr(v_alpha), ptr(v_rec), None, st), "bwd")

        def sort():
            ops.isect_tiles(m2d, radii, dep, 16, 40, 30, tiles_per_gauss=tiles)

        flat_buf = torch.empty(M, dtype=torch.int32, device=dev)

        def sort2():
            ops.isect_bin_sort(m2d, radii, dep, 40, 30, M, None, flat_buf)

        res = {}
        for rnd in range(2):
            for ver in ("1", "0"):
                os.environ["GSX_RASTER_V1"] = ver
                fwd()
                res.setdefault(("fwd", ver), []).append(timed(fwd))
                res.setdefault(("bwd", ver), []).append(timed(bwd))
        os.environ["GSX_RASTER_V1"] = "0"
        for rnd in range(2):
            for sc_ in ("1", "0"):
                os.environ["GSX_BWD_SCALAR"] = sc_
                bwd()
                res.setdefault(("bwd-scalar" if sc_ == "1" else "bwd-readlane", "0"), []).append(timed(bwd))
        os.environ["GSX_BWD_SCALAR"] = "0"
        for k, v in sorted(res.items()):
            print(f"  raster_{k[0]} {'v1' if k[1] == '1' else 'v2'}: median/min us per round = "
                  + ", ".join(f"{a:.1f}/{b:.1f}" for a, b in v))
        os.environ["GSX_SORT_V1"] = "1"
        print(f"  isect_tiles v1 rocPRIM (scan..sort, incl. M read-back): {timed(sort)[0]:.1f} us")
        os.environ["GSX_SORT_V1"] = "0"
        print(f"  isect_bin_sort v2 (diff/offsets/emit/tile-sort, sync-free): {timed(sort2)[0]:.1f} us")
        assert torch.equal(flat_buf, flat), "v2 sort differs from v1"


if __name__ == "__main__":
    main()
