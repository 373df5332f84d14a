"""Render forward + backward time and algorithmic HBM rate at the BASELINE.json configurations (SURVEY.md 8d byte
model: B_fwd = 92 C N + 92 M + 28 P, B_bwd = 136 C N + 40 N + 48 M + 32 P at CH = 5; SH-3 adds N 192 + C N 12 each
way), one GPU.  Configs 2-5; config 4's 8 keyframes are what ONE GPU would render without sharding.

    python tools/bench_configs.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd.rasterization import IsectCapacity, rasterization  # noqa: E402
from gslam_amd.rendering import rasterization as gs_rasterization  # noqa: E402
from gslam_amd.synthetic import make_cameras, make_scene  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def graphed(step):
    """the same step replayed from a HIP graph: what the GPU needs once the ~20 launches of a render and the autograd
    bookkeeping are off the critical path (the small configurations are launch-bound when run eagerly)"""
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
    return g.replay


def run(name, N, C, W, H, sh=False):
    sc = make_scene(N, 0, sh_degree=3 if sh else None)
    viewmats, Ks = make_cameras(C, W, H)
    viewmats, Ks = viewmats.to(dev), Ks.to(dev)
    bg = torch.zeros(C, 3, device=dev)
    state = {}
    cap = IsectCapacity(dev)        # sync-free renders: no read-back of M inside a step (both entry points)
    # fresh leaves per timing mode: autograd pins a leaf's AccumulateGrad node to the stream of its first backward, and
    # the graph is captured on a side stream
    p = {k: v.to(dev).requires_grad_(v.is_floating_point()) for k, v in sc.items()}

    def step():
        for v in p.values():
            v.grad = None
        if sh:
            render, alphas, info = gs_rasterization(p["means"], p["quats"], torch.exp(p["scales"]),
                                                    torch.sigmoid(p["opacities"]), p["sh_coeffs"], viewmats, Ks, W, H,
                                                    sh_degree=3, packed=False, backgrounds=bg, capacity=cap)
            (render.sum() + alphas.sum()).backward()
        else:
            out = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], viewmats, Ks, W, H,
                                packed=False, render_mode="RGB+D", log_uncertainties=p["log_uncertainties"],
                                backgrounds=bg, capacity=cap)
            (out._render.sum() + out.alphas.sum()).backward()

    step()
    assert cap.validate()
    step()
    assert cap.validate()
    state["M"] = cap.last_M
    ms = timed(step, 6 if N >= 2_000_000 else 10)
    try:
        for k in list(p):
            p[k] = p[k].detach().clone().requires_grad_(p[k].is_floating_point())
        torch.cuda.synchronize()
        gms = timed(graphed(step), 6 if N >= 2_000_000 else 20)
        assert cap.validate(), "tile lists overflowed under replay"
    except Exception as e:  # noqa: BLE001 - report the eager number alone
        print(f"graph replay unavailable for {name}: {e!r}", file=sys.stderr)
        gms = None
    best = min(ms, gms) if gms else ms
    M, P, CH = state["M"], C * W * H, 3 if sh else 5
    b = (C * N * 92 + M * (72 + 4 * CH) + P * (4 * CH + 8)) + (C * N * (116 + 4 * CH) + N * 40 + M * (28 + 4 * CH) +
                                                              P * (4 * CH + 12))
    if sh:
        b += 2 * (N * 192 + C * N * 12) + N * 192
    print(json.dumps({"config": name, "gaussians": N, "cameras": C, "size": f"{W}x{H}", "intersections": M,
                      "fwd_bwd_ms": round(ms, 3), "fwd_bwd_ms_graph": round(gms, 3) if gms else None,
                      "algorithmic_MB": round(b / 1e6, 1), "algorithmic_GBps": round(b / best / 1e6, 1),
                      "frac_of_8TBps": round(b / best / 1e6 / 8000.0, 4)}),
          flush=True)
    del p, state
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("2: 100k, 640x480, 1 camera", 100_000, 1, 640, 480)
    run("3: 500k, 640x480, 1 camera (tracking closure)", 500_000, 1, 640, 480)
    run("3: 500k, 640x480, 8 cameras (BA window)", 500_000, 8, 640, 480)
    run("4: 2M, 640x480, 1 camera (one GPU's share of the sharded BA)", 2_000_000, 1, 640, 480)
    run("4: 2M, 640x480, 8 cameras (unsharded)", 2_000_000, 8, 640, 480)
    run("5: 5M SH-3, 1920x1080, 1 camera", 5_000_000, 1, 1920, 1080, sh=True)
