"""Where and when the wavefronts of the tracking closure's rasteriser kernels run (DIAGNOSTIC library built with
-DGSX_WG_TRACE by tools/dbg/wg_trace.sh; its run times are not quotable).  Prints, per kernel: the kernel's span, the
distribution of wavefront durations, how busy the SIMDs were and when each finished - the load balance of the static
workgroup -> CU placement."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def analyse(name, buf, waves_per_wg, work=None):
    a = buf.cpu().numpy().reshape(-1, 4)
    a = a[a[:, 1] > 0]
    t0, t1 = a[:, 0].astype(np.float64) / 100.0, a[:, 1].astype(np.float64) / 100.0      # us
    base = t0.min()
    t0 -= base
    t1 -= base
    hw, xcc = a[:, 2].astype(np.int64), a[:, 3].astype(np.int64) & 15
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    span = t1.max()
    dur = t1 - t0
    key_cu = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    key_simd = key_cu * 4 + simd
    print(f"== {name}: {len(a)} wavefronts, span {span:.1f} us, start spread {t0.max():.1f} us")
    print(f"   wavefront duration: mean {dur.mean():.1f}  p10 {np.percentile(dur, 10):.1f}  p50 {np.percentile(dur, 50):.1f}  "
          f"p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f} us")
    ncu, nsimd = len(np.unique(key_cu)), len(np.unique(key_simd))
    fin = np.array([t1[key_simd == k].max() for k in np.unique(key_simd)])
    cnt = np.array([(key_simd == k).sum() for k in np.unique(key_simd)])
    print(f"   {ncu} CUs / {nsimd} SIMDs used; wavefronts per SIMD: mean {cnt.mean():.2f} min {cnt.min()} max {cnt.max()}")
    print(f"   SIMD finish time: mean {fin.mean():.1f}  p10 {np.percentile(fin, 10):.1f}  p50 {np.percentile(fin, 50):.1f}  "
          f"p90 {np.percentile(fin, 90):.1f}  max {fin.max():.1f} us   -> mean/max = {fin.mean() / fin.max():.2f}")
    fin_cu = np.array([t1[key_cu == k].max() for k in np.unique(key_cu)])
    print(f"   CU finish time:   mean {fin_cu.mean():.1f}  p10 {np.percentile(fin_cu, 10):.1f}  p90 {np.percentile(fin_cu, 90):.1f}"
          f"  max {fin_cu.max():.1f} us   -> mean/max = {fin_cu.mean() / fin_cu.max():.2f}")
    kx = np.array([k // (8 * 2 * 16) for k in np.unique(key_cu)])
    print("   CU finish time by XCD:", [round(float(fin_cu[kx == x].mean()), 1) for x in range(8)])
    nw = np.array([(key_cu == k).sum() // waves_per_wg for k in np.unique(key_cu)])
    for n in np.unique(nw):
        f = fin_cu[nw == n]
        print(f"   CUs with {n} workgroups: finish mean {f.mean():.1f} min {f.min():.1f} max {f.max():.1f}")
    # how many wavefronts are alive over time (fraction of the peak)
    ts = np.linspace(0, span, 11)[1:-1]
    alive = [(int(((t0 <= t) & (t1 > t)).sum())) for t in ts]
    print("   wavefronts alive at 10..90 % of the span:", alive)
    wg = np.arange(len(a)) // waves_per_wg
    first = [(int(xcc[i * waves_per_wg]), int(se[i * waves_per_wg]), int(sh[i * waves_per_wg]), int(cu[i * waves_per_wg]))
             for i in range(0, 40)]
    print("   (xcc, se, sh, cu) of workgroups 0..39:", first)
    kwg = key_cu[::waves_per_wg]
    nwg = len(kwg)
    for stride in (8, 32, 64, 128, 256, 512):
        same = float((kwg[:-stride] == kwg[stride:]).mean())
        print(f"   workgroups i and i+{stride} on the same CU: {same:.3f}")
    grp = {}
    for i, k in enumerate(kwg):
        grp.setdefault(int(k), []).append(i)
    sizes = np.array([len(v) for v in grp.values()])
    print("   workgroups per CU: ", {int(x): int((sizes == x).sum()) for x in np.unique(sizes)})
    mod_ok = sum(1 for v in grp.values() if len(set(i % 256 for i in v)) == 1)
    print(f"   CUs whose workgroups all share i mod 256: {mod_ok} of {len(grp)}")
    print("   a few CUs' workgroup lists:", [v for v in list(grp.values())[:6]])
    if work is not None:
        w = work.cpu().numpy().astype(np.float64)
        wd = np.array([dur[wg == i].max() for i in range(len(w))])
        ok = w > 0
        print(f"   per-workgroup duration vs processed entries: corr {np.corrcoef(w[ok], wd[ok])[0, 1]:.3f}; "
              f"entries mean {w.mean():.0f} p90 {np.percentile(w, 90):.0f} max {w.max():.0f}; "
              f"us per 1000 entries (median) {np.median(wd[ok] / w[ok]) * 1e3:.1f}")
        # what a perfectly balanced static placement of these workgroups over the CUs would reach (LPT on durations)
        order = np.argsort(-wd)
        bins = np.zeros(ncu)
        for i in order:
            bins[np.argmin(bins)] += wd[i]
        per_cu_now = np.array([wd[np.unique(wg[key_cu == k])].sum() for k in np.unique(key_cu)])
        print(f"   sum of workgroup durations per CU now: mean {per_cu_now.mean():.1f} max {per_cu_now.max():.1f}; "
              f"LPT-balanced max {bins.max():.1f}  (ratio {per_cu_now.max() / bins.max():.2f})")


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
    f = frames[9]
    tr.load(f)
    tr.plan.init_optimizer(conf.n_adam_warmup, conf.pose_optim_lr, conf.lbfgs_history, 25)
    st = current_stream_ptr(dev)
    for _ in range(200):                                   # clocks up
        tr.plan.enqueue(st)
    torch.cuda.synchronize()
    T = tr.plan.r.T
    bufs = {k: torch.zeros(T * 4 * 4, dtype=torch.int64, device=dev) for k in ("gsx_raster_fwd_track_loss", "gsx_raster_bwd")}
    import ctypes as C
    setter = lib.gsx_debug_wg_trace
    setter.argtypes, setter.restype = [C.c_int, C.c_void_p], C.c_int
    graphed = "--graph" in sys.argv       # closures replayed back to back from the captured graph instead of one eager closure
    for i, k in enumerate(bufs):
        assert setter(i, bufs[k].data_ptr()) == 0
    r = tr.plan.r

    def traced(tag):
        for k in bufs:
            bufs[k].zero_()
        torch.cuda.synchronize()
        if graphed:
            tr.plan.graph.launch(count=5)
        else:
            tr.plan.enqueue(st)
        torch.cuda.synchronize()
        order = r.launch_order
        tw = r.tile_work.clone() if r.tile_work is not None else None
        for k, name in (("gsx_raster_fwd_track_loss", "forward"), ("gsx_raster_bwd", "backward")):
            w = None
            if tw is not None:
                w = tw[:, 1].float() + r.CHUNK_COST * tw[:, 0].float()
                if order is not None:
                    w = w[order.to(torch.int64)]
            analyse(f"{tag}: raster {name}", bufs[k], 4, w)
            if tw is not None:
                a = bufs[k].cpu().numpy().reshape(-1, 4)
                dur = ((a[:, 1] - a[:, 0]) / 100.0).reshape(-1, 4).max(axis=1)
                o = order.cpu().numpy() if order is not None else np.arange(T)
                np.save(f"gpurun_out/wg_{tag}_{name}.npy", np.stack([o, dur, tw.cpu().numpy()[o, 0], tw.cpu().numpy()[o, 1]], 1))

    if "--b2b" in sys.argv:
        # the same backward launch repeated back to back on one stream: do consecutive launches overlap?
        last = {}
        orig_bwd = lib.gsx_raster_bwd

        def grab(*a):
            last["a"] = a
            return orig_bwd(*a)
        lib.gsx_raster_bwd = grab
        tr.plan.enqueue(st)
        lib.gsx_raster_bwd = orig_bwd
        torch.cuda.synchronize()
        for k in bufs:
            bufs[k].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(300):
            orig_bwd(*last["a"])
        e0.record()
        for _ in range(100):
            orig_bwd(*last["a"])
        e1.record()
        torch.cuda.synchronize()
        print("back to back: %.1f us per launch by events" % (e0.elapsed_time(e1) * 10.0))
        analyse("back-to-back raster backward (last launch)", bufs["gsx_raster_bwd"], 4, None)
        return
    if "--lag" in sys.argv:
        # how fast the tiles' work decorrelates along the closures of a frame
        frames2, _ = bench.make_frames([sequence_param(i) for i in range(3, 6)], W, H, dev,
                                       GaussianSplattingData.from_dict(make_scene(N, 1), dev))
        for fr in frames2:
            tr.load(fr)
            tr.plan.init_optimizer(conf.n_adam_warmup, conf.pose_optim_lr, conf.lbfgs_history, 25)
            ws, poses = [], []
            for k in range(36):
                tr.plan.graph.launch(count=1)
                torch.cuda.synchronize()
                ws.append((r.tile_work[:, 1].float() + 3 * r.tile_work[:, 0].float()).cpu().numpy())
                poses.append(r.viewmats[0].cpu().numpy().copy())
            c1 = [round(float(np.corrcoef(ws[k - 1], ws[k])[0, 1]), 2) for k in range(1, 36)]
            c0 = [round(float(np.corrcoef(ws[0], ws[k])[0, 1]), 2) for k in range(1, 36)]
            cl = [round(float(np.corrcoef(ws[-1], ws[k])[0, 1]), 2) for k in range(0, 35)]
            dp = [round(float(np.abs(poses[k][:3, 3] - poses[k - 1][:3, 3]).max() * 1e3), 2) for k in range(1, 36)]
            print("lag-1 corr:", c1)
            print("corr with closure 0:", c0)
            print("corr with the last closure:", cl)
            print("translation step (mm):", dp)
        return
    if "--frames" in sys.argv:
        # the bench's flow: track() per frame (load, rebalance from the previous frame's last closure, 36 graph launches)
        def spans():
            out = []
            for k in bufs:
                a = bufs[k].cpu().numpy().reshape(-1, 4)
                a = a[a[:, 1] > 0]
                out.append(round(float(a[:, 1].max() - a[:, 0].min()) / 100.0, 1))
            return out
        frames2, _ = bench.make_frames([sequence_param(i) for i in range(3, 12)], W, H, dev,
                                       GaussianSplattingData.from_dict(make_scene(N, 1), dev))
        for i, fr in enumerate(frames2):
            tw0 = r.tile_work.clone()
            tr.track(fr, sync=False)
            torch.cuda.synchronize()
            o = r.balanced_order.cpu().numpy()
            w = (tw0[:, 1].float() + 3 * tw0[:, 0].float()).cpu().numpy()
            w1 = (r.tile_work[:, 1].float() + 3 * r.tile_work[:, 0].float()).cpu().numpy()
            g = np.arange(T) % 256
            s0 = np.array([w[o][g == b].sum() for b in range(256)])
            s1 = np.array([w1[o][g == b].sum() for b in range(256)])
            print(f"frame {i}: last-closure spans fwd/bwd {spans()} us; group sums by the weights used: max/mean {s0.max() / s0.mean():.3f}; "
                  f"by this frame's last closure: max/mean {s1.max() / s1.mean():.3f}; corr(old w, new w) {np.corrcoef(w, w1)[0, 1]:.3f}")
        return
    if r.balanced_order is not None:
        r.balanced_order.copy_(torch.arange(T, dtype=torch.int32, device=dev))
    traced("identity")
    if r.balanced_order is not None:
        r.rebalance(st)
        torch.cuda.synchronize()
        o = r.balanced_order.cpu().numpy()
        assert sorted(o.tolist()) == list(range(T)), "not a permutation"
        for _ in range(100):
            tr.plan.enqueue(st)
        torch.cuda.synchronize()
        traced("balanced")
    return
    off = r.offsets[:T + 1].to(torch.int64)
    last = r.last_ids.view(-1)
    # processed entries per tile: up to the deepest pixel's last entry
    Ht, Wt = (H + 15) // 16, (W + 15) // 16
    lastmax = last.view(Ht, 16, Wt, 16).permute(0, 2, 1, 3).reshape(T, 256).max(dim=1).values.to(torch.int64)
    work = (lastmax + 1 - off[:T]).clamp(min=0)
    listlen = off[1:T + 1] - off[:T]
    print("tile list length: mean %.0f max %d; processed: mean %.0f max %d" % (float(listlen.float().mean()), int(listlen.max()),
                                                                                 float(work.float().mean()), int(work.max())))
    order = r.tile_order
    if order is not None:
        work = work[order.to(torch.int64)]
    analyse("raster forward (+ loss)", bufs["gsx_raster_fwd_track_loss"], 4, work)
    analyse("raster backward (geometry-only)", bufs["gsx_raster_bwd"], 4, work)


if __name__ == "__main__":
    main()
