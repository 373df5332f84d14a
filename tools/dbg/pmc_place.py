import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd import ops
from gslam_amd.synthetic import make_cameras, make_scene
dev = torch.device("cuda:0")
N, C, W, H = (5000000, 1, 1920, 1080) if len(sys.argv) > 1 and sys.argv[1] == "5" else (500000, 8, 640, 480)
sc = {k: v.to(dev) for k, v in make_scene(N, 0).items()}
viewmats, Ks = make_cameras(C, W, H)
radii, m2d, dep, con, _ = ops.fully_fused_projection(sc["means"], None, sc["quats"], torch.exp(sc["scales"]), viewmats.to(dev), Ks.to(dev), W, H)
tw, th = math.ceil(W / 16), math.ceil(H / 16)
tpg = torch.empty(C, N, dtype=torch.int32, device=dev)
from gslam_amd._lib import check, lib, ptr, stream_ptr
check(lib.gsx_isect_count(ptr(m2d), ptr(radii), C * N, tw, th, ptr(tpg), stream_ptr(dev)), "count")
M = int(tpg.sum().item())
buf = torch.empty(int(M * 1.2), dtype=torch.int32, device=dev)
for _ in range(3):
    ops.isect_bin_sort(m2d, radii, dep, tw, th, buf.shape[0], None, buf)
torch.cuda.synchronize()
print("M", M)
