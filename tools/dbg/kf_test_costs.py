"""Backend.to_insert_keyframe (reference backend.py:739-792) piece by piece on an idle GPU: where do its ~20 ms go?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gslam_amd.map import GaussianSplattingData  # noqa: E402
from gslam_amd.synthetic import make_scene  # noqa: E402

dev = torch.device("cuda:0")
N, W, H = 500_000, 640, 480
scene = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
frames, cam = bench.make_frames([0, 1], W, H, dev, scene)
a, b = frames


def t(label, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print(f"{label:50s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms")
    return r


with torch.no_grad():
    pass
out = t("splats([2 cams], render_depth=True) (autograd on)", lambda: scene([a.camera, b.camera], [a.pose, b.pose], render_depth=True))
with torch.no_grad():
    out = t("  the same under no_grad", lambda: scene([a.camera, b.camera], [a.pose, b.pose], render_depth=True))
t("inv(pose) @ pose, norm, .item()", lambda: (torch.linalg.inv(a.pose()) @ b.pose())[:3, 3].pow(2.0).sum().pow(0.5).item())
seen = out.alphas[..., 0] > 0.1
t("seen.any() -> bool", lambda: bool(seen.any()))
t("depthmaps[seen].median()", lambda: out.depthmaps[seen].median())
t("depthmaps[seen].median().item()", lambda: out.depthmaps[seen].median().item())
t("cosine_similarity + bool", lambda: bool(torch.nn.functional.cosine_similarity(a.pose()[:3, 2], b.pose()[:3, 2], dim=0) < 0.9))
