"""Host profile of one backend idle cycle (optimize_map + run_pruning + optimize_poses_lbfgs + sync) on an otherwise idle
GPU at the headline shape: cProfile of 8 cycles, top functions by cumulative time."""
import cProfile
import os
import pstats
import queue
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gslam_amd.backend import Backend, MapConfig  # noqa: E402
from gslam_amd.map import GaussianSplattingData  # noqa: E402
from gslam_amd.synthetic import make_scene  # noqa: E402

dev = torch.device("cuda:0")
N, W, H = 500_000, 640, 480
gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
frames, cam = bench.make_frames(list(range(bench.WINDOW)), W, H, dev, gt)
del gt
be = Backend(MapConfig(device=str(dev)), queue.Queue(), queue.Queue())
be.splats = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
be.initialize_optimizers()
for i, f in enumerate(frames):
    f.index = i
    if i == 0:
        for p in f.pose.parameters():
            p.requires_grad_(False)
    f.exposure_params = f.exposure_params.detach()
    be.keyframes[i] = f
    be.pose_graph[i] = set()
for i in range(bench.WINDOW):
    be.ba.optimizers.add_pose(be.keyframes[i].pose)
be._render_last_keyframe()


def cycle():
    be.pause_map_optim = False
    be.optimize_map()
    be.run_pruning()
    be.optimize_poses_lbfgs()
    be.sync()


for _ in range(3):
    cycle()
torch.cuda.synchronize()
parts = {}
for name in ("optimize_map", "run_pruning", "optimize_poses_lbfgs", "sync"):
    parts[name] = 0.0
t0 = time.perf_counter()
for _ in range(8):
    be.pause_map_optim = False
    for name in parts:
        ta = time.perf_counter()
        getattr(be, name)()
        parts[name] += time.perf_counter() - ta
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"cycle {el / 8 * 1e3:.2f} ms:", {k: round(v / 8 * 1e3, 2) for k, v in parts.items()}, "N", be.splats.means.shape[0])
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    cycle()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
