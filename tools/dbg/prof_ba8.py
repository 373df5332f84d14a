import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd.map import GaussianSplattingData
from gslam_amd.mapping import BundleAdjuster, GraphedBundleAdjuster
from gslam_amd.primitives import Camera, Frame, PoseZhou
from gslam_amd.rasterization import validate
from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
W, H = 640, 480
K = make_intrinsics(W, H).to(dev); cam = Camera(K, H, W)
m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
kf = []
for i in range(8):
    V = make_viewmat(i).to(dev)
    with torch.no_grad():
        img = gt([cam], [PoseZhou(V, is_learnable=False).to(dev)]).rgbs[0].clamp(0, 1).contiguous()
    kf.append(Frame(img=img, timestamp=0.0, camera=cam, pose=PoseZhou(V).to(dev), gt_pose=V, index=i, exposure_params=torch.zeros(2, device=dev)))
ba = BundleAdjuster(m, capturable=True)
gba = GraphedBundleAdjuster(ba, kf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    gba.step()
e1.record(); torch.cuda.synchronize()
print("ms per BA step (C=8):", e0.elapsed_time(e1) / 30, "capacity ok", validate(dev))
