"""Where a wavefront of the fused tracking rasteriser spends its shader cycles (DIAGNOSTIC library built with -DGSX_WG_TRACE
by tools/dbg/phase_trace.sh; the stamps perturb the kernel, its run time is not quotable): every wavefront books the cycles
between consecutive stamps under one of the categories below; printed as mean cycles per wavefront and share of the total."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

CATS = {0: "prologue (until the forward's chunk loop)", 1: "fwd: chunk gather + cull + staging", 2: "fwd: survivor loop",
        3: "loss epilogue + workgroup barrier", 4: "bwd: batch prologue (barrier, ids, zero, barrier)",
        5: "bwd: chunk gather + cull + staging", 6: "bwd: survivor bodies", 7: "bwd: group flush (matrix instructions)",
        8: "bwd: conversion to moments", 9: "bwd: batch flush (barrier + global atomics)"}


def main():
    import bench
    from gslam_amd import _lib
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import current_stream_ptr
    from gslam_amd.synthetic import make_scene, sequence_param
    from gslam_amd.tracking import GraphedTracker, TrackingConfig
    lib = _lib.lib
    dev = torch.device("cuda:0")
    N, W, H = 500_000, 640, 480
    gt_scene = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
    m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    frames, cam = bench.make_frames(list(range(8)) + [sequence_param(i) for i in range(3)], W, H, dev, gt_scene)
    del gt_scene
    conf = TrackingConfig()
    tr = GraphedTracker(m.no_grad_clone(), cam, conf)
    tr.track(frames[8])
    tr.load(frames[9])
    tr.plan.init_optimizer(conf.n_adam_warmup, conf.pose_optim_lr, conf.lbfgs_history, 25)
    st = current_stream_ptr(dev)
    for _ in range(100):
        tr.plan.enqueue(st)
    torch.cuda.synchronize()
    T = tr.plan.r.T
    buf = torch.zeros(T * 4 * 16, dtype=torch.int64, device=dev)
    setter = lib.gsx_debug_phase_trace
    setter.argtypes, setter.restype = [C.c_void_p], C.c_int
    assert setter(buf.data_ptr()) == 0
    tr.plan.enqueue(st)
    torch.cuda.synchronize()
    assert setter(None) == 0
    a = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    a = a[a[:, :10].sum(axis=1) > 0]
    tot = a[:, :10].sum(axis=1)
    print(f"{len(a)} wavefronts; cycles per wavefront: mean {tot.mean():.0f}  p10 {np.percentile(tot, 10):.0f}  p90 {np.percentile(tot, 90):.0f}"
          f"  max {tot.max():.0f}   (the diagnostic build; ~{tot.mean() / 2.4e3:.1f} us at 2.4 GHz)")
    for k, name in CATS.items():
        print(f"  {name:52s} {a[:, k].mean():9.0f} cycles  {100 * a[:, k].mean() / tot.mean():5.1f} %")
    print(f"  survivors per wavefront: fwd {a[:, 10].mean():.1f}  bwd {a[:, 11].mean():.1f};  chunks: fwd {a[:, 12].mean():.2f}  bwd {a[:, 13].mean():.2f}")
    print(f"  cycles per survivor: fwd loop {a[:, 2].sum() / max(a[:, 10].sum(), 1):.0f}  bwd bodies {a[:, 6].sum() / max(a[:, 11].sum(), 1):.0f}"
          f"  bwd flush {a[:, 7].sum() / max(a[:, 11].sum(), 1):.0f};  per chunk: fwd {a[:, 1].sum() / max(a[:, 12].sum(), 1):.0f}  bwd {a[:, 5].sum() / max(a[:, 13].sum(), 1):.0f}")


if __name__ == "__main__":
    main()
