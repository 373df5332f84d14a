"""debug: per-pixel outputs of the sort-inside launch against the stand-alone sort at a pose jump"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else '.')
from gslam_amd._lib import check, lib
from gslam_amd.map import GaussianSplattingData
from gslam_amd.plan import TrackClosure, current_stream_ptr, _p, _COMPACT
from gslam_amd.primitives import Camera
from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat

dev = torch.device('cuda:0')
W, H = 640, 480
sc = make_scene(80000, 6)
sc["scales"] = sc["scales"] + 0.4
splats = GaussianSplattingData.from_dict(sc, dev)
cam = Camera(make_intrinsics(W, H).to(dev), H, W)
img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
st = current_stream_ptr(dev)
ref = TrackClosure(splats, cam, defer_sort=False)
new = TrackClosure(splats, cam, defer_sort=True, **({'near_place': False} if len(sys.argv) <= 1 else {}))
for c in (ref, new):
    c.load(make_viewmat(2.0).to(dev), img, torch.tensor([0.02, -0.01], device=dev))
    c.r.probe()


def run(c, sorting):
    r = c.r
    r.v_rec.zero_()
    al = torch.zeros(1, H, W, 1, device=dev); li = torch.zeros(1, H, W, dtype=torch.int32, device=dev)
    vr = torch.zeros(1, H, W, 4, device=dev)
    if sorting:
        r._front(st, defer_sort=True)
        lay = (C.c_int64 * 3)()
        check(lib.gsx_front_keys(r.N, r.C, r.tile_w, r.tile_h, r.capacity, _COMPACT, lay), "k")
        base = r.isect_ws.data_ptr()
        check(lib.gsx_raster_track_fused_sorting(
            _p(r.rec), _p(r.backgrounds), _p(r.offsets), _p(r.flat), r.capacity, 1, r.C, r.W, r.H,
            _p(c.img), _p(c.exposure), 1.0 / (H * W), _p(al), _p(li), _p(vr), _p(c.loss_rows), _p(r.v_rec), _p(r.launch_order),
            _p(r.tile_work), base + int(lay[0]), base + int(lay[1]), int(lay[2]), _p(r.tile_cut),
            float(r.cut_margin), _p(r.tile_near), _p(r.sort_stats), st), "s")
    else:
        r._front(st)
        check(lib.gsx_raster_track_fused(_p(r.rec), _p(r.backgrounds), _p(r.offsets), _p(r.flat), r.capacity,
                                         1, r.C, r.W, r.H, _p(c.img), _p(c.exposure), 1.0 / (H * W), _p(al), _p(li), _p(vr),
                                         _p(c.loss_rows), _p(r.v_rec), _p(r.launch_order), _p(r.tile_work), st), "f")
    torch.cuda.synchronize()
    assert r.check_capacity()
    return c.loss_rows.clone(), al, li, vr, r.offsets[:r.T + 1].clone(), r.flat.clone()


for k in range(2):
    a = run(ref, False); b = run(new, True)
    print("closure", k, "loss diff", float((a[0] - b[0]).abs().max()), "alpha diff", float((a[1] - b[1]).abs().max()))
V2 = make_viewmat(2.0)
cth, sth = float(np.cos(0.06)), float(np.sin(0.06))
Rj = torch.tensor([[cth, 0.0, sth, 0.0], [0.0, 1.0, 0.0, 0.0], [-sth, 0.0, cth, 0.0], [0.0, 0.0, 0.0, 1.0]])
V2 = Rj @ V2
V2[0, 3] += 0.05
for c in (ref, new):
    c.r.viewmats[0].copy_(V2.to(dev))
for name in ("jump", "again"):
    a = run(ref, False); b = run(new, True)
    off = a[4].cpu().numpy()
    near = new.r.tile_near.cpu().numpy()
    d = (a[0] - b[0]).abs().max(dim=1).values.cpu().numpy()
    print(name, "loss diff", d.max(), "alpha diff", float((a[1] - b[1]).abs().max()), "last differ", int((a[2] != b[2]).sum()),
          "v_render diff", float((a[3] - b[3]).abs().max()))
    bad = np.nonzero(d > 1e-6 * float(a[0].abs().max()))[0]
    A, B = a[1][0, :, :, 0].cpu().numpy(), b[1][0, :, :, 0].cpu().numpy()
    LA, LB = a[2][0].cpu().numpy(), b[2][0].cpu().numpy()
    for t in bad[:6]:
        ty, tx = divmod(int(t), 40)
        sl = (slice(ty * 16, ty * 16 + 16), slice(tx * 16, tx * 16 + 16))
        da = np.abs(A[sl] - B[sl])
        ys, xs = np.nonzero((da > 0) | (LA[sl] != LB[sl]))
        print(" tile", t, "start", off[t], "near", near[t], "size", off[t + 1] - off[t], "pixels differing", len(ys))
        for y, x in list(zip(ys, xs))[:4]:
            print("    px", y, x, "alpha", A[sl][y, x], B[sl][y, x], "last - start", LA[sl][y, x] - off[t], LB[sl][y, x] - off[t])
