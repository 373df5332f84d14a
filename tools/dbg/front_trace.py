"""When the workgroups of the two front kernels of a tracking closure (front_project_kernel, front_place_kernel) reach their
phase boundaries (DIAGNOSTIC library built with -DGSX_WG_TRACE by tools/dbg/front_trace.sh): thread 0 of every workgroup
stamps s_memrealtime (100 MHz); printed in microseconds after the kernel's first stamp."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

NAMES = [["entry", "LDS zeroed", "records loaded, projected, compacted", "rows + instance records written (issue)",
          "rectangles counted", "workgroup done counting", "phase 2: covariance of the survivor ready", "phase 1 done (cull + compaction)"],
         ["entry", "tile totals scanned", "cursors of the stripe ready", "placed (issue)", "first round's list built"]]


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
    buf = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device=dev)
    setter = lib.gsx_debug_front_trace
    setter.argtypes, setter.restype = [C.c_void_p], C.c_int
    assert setter(buf.data_ptr()) == 0
    if os.environ.get("FRONT_ONLY"):                     # the two front kernels alone (no rasteriser behind them)
        for _ in range(3):
            tr.plan.r._front(st, defer_sort=True)
        torch.cuda.synchronize()
        buf.zero_()
        tr.plan.r._front(st, defer_sort=True)
    else:
        tr.plan.enqueue(st)
    torch.cuda.synchronize()
    assert setter(None) == 0
    a = buf.cpu().numpy().reshape(2, 4096, 8).astype(np.float64)
    for k, kern in enumerate(("front_project_kernel", "front_place_kernel")):
        if k == 1 and tr.plan.r.row_keys:
            continue
        rows = a[k][a[k][:, 0] > 0]
        t0 = rows[:, 0].min()
        print(f"{kern}: {len(rows)} workgroups stamped; microseconds after the first entry")
        for j, name in sorted(enumerate(NAMES[k]), key=lambda x: {7: 1.5, 6: 1.7}.get(x[0], x[0]) if k == 0 else {4: 2.5}.get(x[0], x[0])):
            v = rows[:, j]
            v = (v[v > 0] - t0) / 100.0
            if len(v):
                print(f"  {name:45s} n={len(v):5d}  min {v.min():6.2f}  p25 {np.percentile(v, 25):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
    if tr.plan.r.row_keys:
        # row keys: no placement launch; the projection workgroups stamp their tail on the second kernel's rows
        rows0 = a[0][a[0][:, 0] > 0]
        t0 = rows0[:, 0].min()
        tail = a[1][:len(rows0)]
        for j, name in enumerate(("row-keys tail entered", "row scanned, words written", "keys placed (issue)")):
            v = (tail[:, j][tail[:, j] > 0] - t0) / 100.0
            print(f"  {name:45s} n={len(v):5d}  min {v.min():6.2f}  p25 {np.percentile(v, 25):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
        return
    ns = a[1][:len(a[0][a[0][:, 0] > 0]), 4]
    print(f"survivors of the cull per projection workgroup: min {ns.min():.0f}  p10 {np.percentile(ns, 10):.0f}  p50 {np.percentile(ns, 50):.0f}"
          f"  p90 {np.percentile(ns, 90):.0f}  max {ns.max():.0f}  sum {ns.sum():.0f};  workgroups with a second trip: {(ns > 1024).sum()}")
    rows = a[0][a[0][:, 0] > 0]
    done = (rows[:, 5] - rows[:, 0].min()) / 100.0
    for lo, hi in ((0, 1024), (1024, 4096)):
        sel = (ns > lo) & (ns <= hi)
        if sel.any():
            print(f"  workgroups with {lo} < survivors <= {hi}: {sel.sum()}, done counting at p50 {np.percentile(done[sel], 50):.2f}  max {done[sel].max():.2f} us")
    pr = a[1][a[1][:, 0] > 0]
    e = (pr[:, 0] - pr[:, 0].min()) / 100.0
    d = (pr[:, 3] - pr[:, 0]) / 100.0
    late = e > 2.0
    print(f"placement workgroups entering later than 2 us: {late.sum()} of {len(pr)}; their run time p50 {np.percentile(d[late], 50) if late.any() else 0:.2f} us;"
          f" the others: run time p50 {np.percentile(d[~late], 50):.2f}  p90 {np.percentile(d[~late], 90):.2f}  max {d[~late].max():.2f},"
          f" last one done at {((pr[:, 3] - pr[:, 0].min()) / 100.0)[~late].max():.2f} us")
    idx = np.nonzero(a[1][:, 0] > 0)[0]
    print("  late workgroups (block index):", idx[late][:40])
    n_cand = tr.plan.r.candidate_stats() if tr.plan.r.candidates else None
    print("candidates:", n_cand, " rows:", tr.plan.r.R if hasattr(tr.plan.r, "R") else "?")


if __name__ == "__main__":
    main()
