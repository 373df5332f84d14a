"""Distribution of the tile-list lengths of BASELINE.json configs[4] (5 M Gaussians, 1920x1080) and configs[3] (2 M x 8 cameras,
640x480): what the per-tile sort kernels have to deal with."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd.rendering import rasterization as gs_rasterization  # noqa: E402
from gslam_amd.synthetic import make_cameras, make_scene  # noqa: E402

dev = torch.device("cuda:0")
for name, N, C, W, H in (("configs[4]", 5_000_000, 1, 1920, 1080), ("configs[3]", 2_000_000, 8, 640, 480)):
    sc = make_scene(N, 0)
    viewmats, Ks = make_cameras(C, W, H)
    with torch.no_grad():
        p = {k: v.to(dev) for k, v in sc.items()}
        render, alphas, info = gs_rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                                torch.sigmoid(p["colors"]), viewmats.to(dev), Ks.to(dev), W, H, packed=False)
    off = info["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
    M = int(info["flatten_ids"].numel())
    sizes = np.diff(np.concatenate([off, [M]]))
    q = np.percentile(sizes, [0, 10, 25, 50, 75, 90, 99, 100]).astype(int)
    print(f"{name}: M = {M}, tiles = {len(sizes)}, keys per tile: min {q[0]} p10 {q[1]} p25 {q[2]} p50 {q[3]} p75 {q[4]} p90 {q[5]} p99 {q[6]} max {q[7]};"
          f" tiles over 2048 / 4032 / 9152 keys: {(sizes > 2048).mean():.2f} / {(sizes > 4032).mean():.2f} / {(sizes > 9152).mean():.2f};"
          f" keys in them: {sizes[sizes > 2048].sum() / M:.2f} / {sizes[sizes > 4032].sum() / M:.2f} / {sizes[sizes > 9152].sum() / M:.2f}")
    del p, render, alphas, info
    torch.cuda.empty_cache()
