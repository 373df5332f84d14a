"""cProfile of the eager (non-graph) BA step: where the host time of ~20 launches goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import torch
sys.argv = ["bench.py"]
import bench
from gslam_amd.map import GaussianSplattingData
from gslam_amd.mapping import BundleAdjuster, MapConfig
from gslam_amd.synthetic import make_scene
dev = torch.device("cuda:0")
N, W, H = 100_000, 640, 480
splats = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
gt_scene = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
window = bench.make_window(1, W, H, dev, gt_scene, {0})
ba = BundleAdjuster(splats, MapConfig(), capturable=True)
for _ in range(20):
    ba.step(window)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    ba.step(window)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    ba.step(window)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
