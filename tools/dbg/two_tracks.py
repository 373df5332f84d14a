import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd.map import GaussianSplattingData
from gslam_amd.primitives import Camera, Frame, PoseZhou
from gslam_amd.rasterization import validate
from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
from gslam_amd.tracking import GraphedTracker, TrackingConfig
dev = torch.device("cuda:0")
W, H, N = 640, 480, 20000
K = make_intrinsics(W, H).to(dev)
cam = Camera(K, H, W)
m = GaussianSplattingData.from_dict(make_scene(N, 0), dev).no_grad_clone()
def mk(i):
    V = make_viewmat(i).to(dev)
    return Frame(img=torch.rand(H, W, 3, device=dev), timestamp=0.0, camera=cam, pose=PoseZhou(V).to(dev), gt_pose=V,
                 index=i, exposure_params=torch.zeros(2, device=dev))
tr = GraphedTracker(m, cam, TrackingConfig())
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
print("t1", tr.track(mk(0)), flush=True)
torch.cuda.synchronize()
if mode == "validate":
    validate(dev)
print("t2", tr.track(mk(1)), flush=True)
torch.cuda.synchronize()
print("t3", tr.track(mk(2)), flush=True)
torch.cuda.synchronize()
print("DONE")
