"""the configs[1] extra of bench.py alone: 100 k Gaussians, one keyframe, full BA step (for profiling)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

r = bench.run_ba(torch.device("cuda:0"), 0, 1, 100_000, 640, 480, 1, 200, 20)
print(round(r["ms_per_iter"], 4), "ms per BA step")
