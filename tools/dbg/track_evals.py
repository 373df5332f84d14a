"""how many closure evaluations does the device optimiser actually use per frame on the bench sequence?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gslam_amd.map import GaussianSplattingData
from gslam_amd.synthetic import make_scene
from gslam_amd.tracking import GraphedTracker, TrackingConfig
dev = torch.device("cuda:0")
N, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 500000, 640, 480
same = len(sys.argv) > 2 and sys.argv[2] == "same"       # track against the map the images were rendered from
gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
m = gt if same else GaussianSplattingData.from_dict(make_scene(N, 0), dev)
frames, cam = bench.make_frames(range(8, 20), W, H, dev, gt)
tr = GraphedTracker(m.no_grad_clone(), cam, TrackingConfig())
prev = None
for i, f in enumerate(frames):
    if i > 0:
        f.pose.Rt.copy_(frames[i - 1].pose().detach())    # start from the previous frame's result, as the frontend does
    loss, n = tr.track(f)
    rep = tr._report.cpu().tolist()
    print(f"frame {f.index}: evals={n} lbfgs_iters={int(rep[2])} stop_reason={int(rep[3])} loss={loss:.5f}")
