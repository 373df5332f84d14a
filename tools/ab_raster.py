"""Interleaved A/B timing of render stages in ONE process (cdna guide rule 24): raster kernel generations
(GSX_RASTER=1..5), rocPRIM sort vs tile-binned sort.
usage: python tools/ab_raster.py [N ...] [C=1 C=8]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import ops  # noqa: E402
from gslam_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402
from gslam_amd.rasterization import rasterization, validate  # noqa: E402
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


def run(N, C, W=640, H=480):
    tw, th = (W + 15) // 16, (H + 15) // 16
    sc = {k: v.to(dev) for k, v in make_scene(N, 0).items()}
    viewmats, Ks = make_cameras(C, W, H)
    viewmats, Ks = viewmats.to(dev), Ks.to(dev)

    def render():
        return rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats, Ks, W, H,
                             packed=False, render_mode="RGB+D", log_uncertainties=sc["log_uncertainties"],
                             backgrounds=torch.zeros(C, 3, device=dev))

    with torch.no_grad():
        out = render()
        if not validate(dev):
            out = render()
            assert validate(dev)
    M = out.flatten_ids.shape[0]
    print(f"N={N} C={C} {W}x{H} M={M} visible={(out.radii > 0).sum().item()}")
    radii, m2d, dep, con, _, rec, tiles, _ = ops._Projection.apply(
        sc["means"], sc["quats"], sc["scales"], viewmats, Ks, sc["opacities"], sc["colors"], sc["log_uncertainties"], W,
        H, 0.3, 0.01, 1e10, 0.0, False, 1 | 2 | 4, True, True)
    off, flat = out.isect_offsets.contiguous(), out.flatten_ids.contiguous()
    bg = torch.zeros(C, 5, device=dev)
    bg[:, 4] = 2.718281828
    render_t = torch.empty(C, H, W, 5, device=dev)
    alphas = torch.empty(C, H, W, 1, device=dev)
    last = torch.empty(C, H, W, dtype=torch.int32, device=dev)
    nt = torch.zeros(C, N, dtype=torch.int32, device=dev)
    v_render = torch.randn(C, H, W, 5, device=dev)
    v_alpha = torch.randn(C, H, W, 1, device=dev)
    v_rec = torch.zeros(C, N, 12, device=dev)
    st = stream_ptr(dev)

    cur_order = [None]                                      # launch order of the tiles (None = spatial)

    def fwd():
        check(lib.gsx_raster_fwd(ptr(rec), 5, ptr(bg), ptr(off), ptr(flat), M, 0, C, W, H, tw, th, 0.5, ptr(render_t),
                                 ptr(alphas), ptr(last), ptr(nt), ptr(cur_order[0]), st), "fwd")

    def bwd():
        check(lib.gsx_raster_bwd(ptr(rec), 5, ptr(bg), ptr(off), ptr(flat), M, 0, C, W, H, tw, th, ptr(alphas),
                                 ptr(last), ptr(v_render), ptr(v_alpha), ptr(v_rec), None, ptr(cur_order[0]), 0, st), "bwd")

    def sort1():
        ops.isect_tiles(m2d, radii, dep, 16, tw, th, tiles_per_gauss=tiles)

    flat_buf = torch.empty(M, dtype=torch.int32, device=dev)

    def sort2():
        ops.isect_bin_sort(m2d, radii, dep, tw, th, M, None, flat_buf)

    res = {}
    for _ in range(2):
        for ver in ("2", "4", "5"):
            os.environ["GSX_RASTER"] = ver
            fwd()
            res.setdefault(("fwd", ver), []).append(timed(fwd))
            res.setdefault(("bwd", ver), []).append(timed(bwd))
    os.environ["GSX_RASTER"] = "4"
    for mode in ("1", "2", "4", "5", "6"):
        os.environ["GSX_BWD_MODE"] = mode
        res[("bwd", "4/mode" + mode)] = [timed(bwd), timed(bwd)]
    outs = {}
    for mode in ("1", "2", "3", "4", "5", "6"):             # same numbers from every accumulation mode
        os.environ["GSX_BWD_MODE"] = mode
        v_rec.zero_()
        bwd()
        outs[mode] = v_rec.clone()
    for mode in ("2", "3", "4", "5", "6"):
        err = (outs[mode] - outs["1"]).abs().max().item() / (outs["1"].abs().max().item() + 1e-20)
        print(f"  bwd mode {mode} vs mode 1: max rel err {err:.2e}")
    os.environ.pop("GSX_BWD_MODE", None)
    os.environ.pop("GSX_RASTER", None)
    # heaviest tiles first: launch order = tiles sorted by descending list length
    flat_off = torch.cat([off.reshape(-1).long(), torch.tensor([M], device=dev)])
    counts = flat_off[1:] - flat_off[:-1]
    order = torch.argsort(counts, descending=True).to(torch.int32).contiguous()
    dev_order = torch.empty(C * tw * th, dtype=torch.int32, device=dev)       # what the library itself produces
    ops.isect_bin_sort(m2d, radii, dep, tw, th, M, None, torch.empty(M, dtype=torch.int32, device=dev),
                       tile_order=dev_order)
    assert torch.equal(torch.sort(dev_order.long())[0], torch.arange(C * tw * th, device=dev)), "not a permutation"
    for name, o in (("spatial", None), ("exact sort", order), ("library buckets", dev_order), ("spatial", None),
                    ("exact sort", order), ("library buckets", dev_order)):
        cur_order[0] = o
        fwd()
        a, b = timed(fwd), timed(bwd)
        print(f"  tile order {name}: fwd {a[0]:.1f} us, bwd {b[0]:.1f} us")
    cur_order[0] = None
    fwd()
    res[("fwd", "auto")] = [timed(fwd)]
    res[("bwd", "auto")] = [timed(bwd)]
    for k, v in sorted(res.items()):
        print(f"  raster_{k[0]} v{k[1]}: median/min us = " + ", ".join(f"{a:.1f}/{b:.1f}" for a, b in v))
    os.environ["GSX_SORT_V1"] = "1"
    print(f"  isect_tiles v1 rocPRIM (scan..sort, incl. M read-back): {timed(sort1)[0]:.1f} us")
    os.environ["GSX_SORT_V1"] = "0"
    print(f"  isect_bin_sort v2 (diff/offsets/emit/tile-sort, sync-free): {timed(sort2)[0]:.1f} us")
    assert torch.equal(flat_buf, flat), "v2 sort differs from v1"


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("C=")]
    cams = [int(a[2:]) for a in sys.argv[1:] if a.startswith("C=")] or [1]
    for N in [int(a) for a in args] or [100_000, 500_000]:
        for C in cams:
            run(N, C)
