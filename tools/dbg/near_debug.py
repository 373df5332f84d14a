"""debug: sort-inside-the-rasteriser at a pose jump, against the stand-alone sort (which tiles differ, and how)"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[2] if len(sys.argv) > 2 else '.')
from gslam_amd.map import GaussianSplattingData
from gslam_amd.plan import TrackClosure, current_stream_ptr
from gslam_amd.primitives import Camera
from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat

near_place = len(sys.argv) > 1 and sys.argv[1] == '1'
dev = torch.device('cuda:0')
W, H = 640, 480
sc = make_scene(80000, 6)
sc["scales"] = sc["scales"] + 0.4
splats = GaussianSplattingData.from_dict(sc, dev)
cam = Camera(make_intrinsics(W, H).to(dev), H, W)
img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
st = current_stream_ptr(dev)
ref = TrackClosure(splats, cam, defer_sort=False)
new = TrackClosure(splats, cam, defer_sort=True, **({'near_place': near_place} if len(sys.argv) <= 2 else {}))
for c in (ref, new):
    c.load(make_viewmat(2.0).to(dev), img, torch.tensor([0.02, -0.01], device=dev))
    c.r.probe()


def run(c):
    tl = (c.img, c.exposure, 1.0 / (H * W), c.loss_rows)
    c.r.v_rec.zero_()
    c.r.forward_track_fused(st, tl)
    torch.cuda.synchronize()
    v = c.r.v_rec.clone()
    c.r.backward(st, rasterised=True)
    torch.cuda.synchronize()
    assert c.r.check_capacity()
    return c.loss_rows.clone(), v, c.r.flat[:c.r.last_M].clone(), c.r.offsets[:c.r.T + 1].clone()


for k in range(2):
    a = run(ref); b = run(new)
    print("closure", k, "loss diff", float((a[0] - b[0]).abs().max()), "stats", new.r.sort_stats.tolist())
V2 = make_viewmat(2.0)
cth, sth = float(np.cos(0.06)), float(np.sin(0.06))
Rj = torch.tensor([[cth, 0.0, sth, 0.0], [0.0, 1.0, 0.0, 0.0], [-sth, 0.0, cth, 0.0], [0.0, 0.0, 0.0, 1.0]])
V2 = Rj @ V2
V2[0, 3] += 0.05
for c in (ref, new):
    c.r.viewmats[0].copy_(V2.to(dev))
cuts = new.r.tile_cut.clone().cpu().numpy().view(np.float32)
new.r.sort_stats.zero_()
a = run(ref); b = run(new)
d = (a[0] - b[0]).abs().max(dim=1).values.cpu().numpy()
off = a[3].cpu().numpy(); sizes = off[1:] - off[:-1]
near = new.r.tile_near.cpu().numpy()
print("jump: loss diff", d.max(), "stats", new.r.sort_stats.tolist(), "offsets equal", bool(torch.equal(a[3], b[3])))
bad = np.nonzero(d > 1e-6 * float(a[0].abs().max()))[0]
print(len(bad), "tiles differ")
fa, fb = a[2].cpu().numpy(), b[2].cpu().numpy()
for t in bad[:12]:
    lo, n = off[t], near[t]
    print(" tile", t, "size", sizes[t], "near", n, "cut", cuts[t], "prefix equal", bool((fa[lo:lo + n] == fb[lo:lo + n]).all()),
          "loss", a[0][t].tolist()[:1], b[0][t].tolist()[:1],
          "placed", (int(new.r.tile_placed[t]) if near_place else -1))
import ctypes as C
from gslam_amd._lib import lib as _l
lay = (C.c_int64 * 3)()
_l.gsx_front_keys(new.r.N, new.r.C, new.r.tile_w, new.r.tile_h, new.r.capacity, 32, lay)
M = int(off[-1])
ks = new.r.isect_ws[int(lay[1]):int(lay[1]) + 8 * M].view(torch.int64).cpu().numpy()
dep = (ks >> 32).astype(np.uint32).view(np.float32)
nf = []
for t in bad[:12]:
    lo, n = off[t], near[t]
    d = dep[lo:lo + n]
    print(" tile", t, "near", n, "of which <= cut", int((d <= cuts[t]).sum()), "sorted ok", bool((np.diff(d) >= 0).all()), "last depths", d[-3:].tolist())
# the same pose again (cut-offs of THIS pose now), and with no cut-offs at all
b2 = run(new)
print("again: loss diff", float((a[0] - b2[0]).abs().max()), new.r.sort_stats.tolist())
new.r.reset_cuts()
b3 = run(new)
print("no cuts: loss diff", float((a[0] - b3[0]).abs().max()), new.r.sort_stats.tolist())
