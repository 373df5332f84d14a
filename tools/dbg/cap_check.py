import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from gslam_amd.map import GaussianSplattingData
from gslam_amd.mapping import BundleAdjuster, MapConfig
from gslam_amd.synthetic import make_scene
dev = torch.device("cuda:0")
N, W, H = 500_000, 640, 480
gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
frames, cam = bench.make_frames(list(range(bench.WINDOW)), W, H, dev, gt)
del gt
m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
ba = BundleAdjuster(m, MapConfig(), capturable=True)
plan = ba.plan(frames)
plan.prepare()
r = plan.r
print("capacity", r.capacity, "T", r.T, "per tile", r.capacity / r.T, "last_M", r.last_M, "tiles sum", int(r.tiles.sum().item()))
