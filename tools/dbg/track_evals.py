"""How many closure evaluations does the tracking optimiser actually USE per frame on the bench sequence (10 Adam + strong-Wolfe
L-BFGS with max_eval 25)?  After TO_PHASE_DONE the remaining closures of the 36-closure graph change nothing."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gslam_amd.map import GaussianSplattingData  # noqa: E402
from gslam_amd.synthetic import make_scene, sequence_param  # noqa: E402
from gslam_amd.tracking import GraphedTracker, TrackingConfig  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    N, W, H = 500_000, 640, 480
    gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
    n = int(os.environ.get("FRAMES", 40))
    frames, cam = bench.make_frames([sequence_param(i) for i in range(n)], W, H, dev, gt)
    del gt
    m = GaussianSplattingData.from_dict(make_scene(N, 0), dev).no_grad_clone()
    tr = GraphedTracker(m, cam, TrackingConfig(), device_optimizer=True, max_eval=bench.MAX_EVAL)
    hist = collections.Counter()
    for f in frames:
        tr.track(f, sync=False)
        rep = tr.plan.read_report().cpu().tolist()
        hist[(int(rep[1]), int(rep[2]), int(rep[3]), int(rep[0]))] += 1
    print("(total_evals, lbfgs_iterations, stop_reason, phase) -> frames")
    for k, v in sorted(hist.items()):
        print(k, v)
    del tr


main()
