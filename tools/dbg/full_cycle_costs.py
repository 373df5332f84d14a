"""Where the host + device time of the full mapping cycle goes that is NOT steady-state replay: wall time inside plan
construction / prepare (probe, warm-up, capture) of the BA step, the pose refiner and the tracker, per call."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gslam_amd import mapping, plan, tracking  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])


def timed(obj, name, label):
    orig = getattr(obj, name)

    def wrapper(*a, **k):
        torch.cuda.current_stream().synchronize()          # this thread's stream only: the other half keeps running
        t0 = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.current_stream().synchronize()
        acc[label][0] += 1
        acc[label][1] += time.perf_counter() - t0
        return r
    setattr(obj, name, wrapper)


timed(plan.MappingStep, "__init__", "MappingStep.__init__")
timed(plan.MappingStep, "prepare", "MappingStep.prepare")
timed(plan.WindowClosure, "__init__", "WindowClosure.__init__")
timed(plan.WindowClosure, "prepare", "WindowClosure.prepare")
timed(plan.TrackClosure, "__init__", "TrackClosure.__init__")
timed(plan.TrackClosure, "prepare", "TrackClosure.prepare")
from gslam_amd import backend as be_mod  # noqa: E402
timed(be_mod.Backend, "optimize_map", "Backend.optimize_map (whole)")
timed(be_mod.Backend, "run_pruning", "Backend.run_pruning (whole)")
timed(be_mod.Backend, "optimize_poses_lbfgs", "Backend.optimize_poses_lbfgs (whole)")
timed(be_mod.Backend, "sync", "Backend.sync (whole)")
timed(be_mod.Backend, "to_insert_keyframe", "Backend.to_insert_keyframe (whole)")
dev = torch.device("cuda:0")
res = bench.run_full_cycle(dev, 500_000, 640, 480, steps=40, warmup=10)
print({k: res[k] for k in ("frames_per_s", "ms_per_frame", "mapping_cycles", "tracker_recaptures_in_timed_region")})
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:45s} calls {n:4d}  total {t * 1e3:8.1f} ms  per call {t / n * 1e3:7.2f} ms")
