"""Workload for rocprofv3: the tracking closure (and optionally BA iterations) of the headline configuration, nothing else.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_X -o X -- python3 tools/prof_closure.py [--frames 5]
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ... -- python3 tools/prof_closure.py --eager --frames 1
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--ba", type=int, default=0, help="BA iterations over an 8-keyframe window after the tracking")
    ap.add_argument("--ba-window", type=int, default=8, help="keyframes in the BA window (1 = BASELINE.json configs[1])")
    ap.add_argument("--eager", action="store_true", help="issue the launches eagerly (counter passes cannot attribute graph nodes)")
    ap.add_argument("--front", type=int, default=-1, help="1 / 0 force the fused front on / off")
    ap.add_argument("--ba-front", type=int, default=-1, help="1 / 0 force the fused front of the BA plan on / off")
    ap.add_argument("--no-balance", action="store_true", help="identity launch order of the rasteriser kernels (A/B)")
    ap.add_argument("--order-per-tile", type=int, default=None, help="RenderPlan.ORDER_MAX_PER_TILE (A/B)")
    ap.add_argument("--chunk-cost", type=float, default=None, help="RenderPlan.CHUNK_COST (A/B)")
    ap.add_argument("--light-rate", type=float, default=None, help="RenderPlan.LIGHT_RATE (A/B)")
    ap.add_argument("--cut-margin", type=float, default=None, help="RenderPlan.CUT_MARGIN of the in-rasteriser tile sort (A/B)")
    ap.add_argument("--no-defer-sort", action="store_true", help="stand-alone tile sort launch (A/B)")
    ap.add_argument("--no-row-keys", action="store_true", help="count matrix scan + placement launch instead of row keys (A/B)")
    ap.add_argument("--no-tile-exact", action="store_true", help="every tile of the 3-sigma square listed (A/B of GSX_PROJ_TILE_EXACT)")
    ap.add_argument("--merge-tail", action="store_true", help="the closure's tail inside the pose backward launch (A/B; off by default)")
    ap.add_argument("--no-near", action="store_true", help="every key placed (round-4 placement), sort inside the rasteriser (A/B)")
    ap.add_argument("--near-margin", type=float, default=None, help="RenderPlan.NEAR_MARGIN of the near placement (A/B)")
    args = ap.parse_args()
    import bench
    if args.order_per_tile is not None:
        import gslam_amd.plan as P2
        P2.RenderPlan.ORDER_MAX_PER_TILE = args.order_per_tile
    if args.chunk_cost is not None or args.light_rate is not None:
        import gslam_amd.plan as P5
        if args.chunk_cost is not None:
            P5.RenderPlan.CHUNK_COST = args.chunk_cost
        if args.light_rate is not None:
            P5.RenderPlan.LIGHT_RATE = args.light_rate
    if args.cut_margin is not None:
        import gslam_amd.plan as P3
        P3.RenderPlan.CUT_MARGIN = args.cut_margin
    if args.no_row_keys:
        import gslam_amd.plan as P9
        P9.RenderPlan.enable_row_keys = lambda self: False
    if args.no_tile_exact:
        import gslam_amd.plan as P10
        P10.RenderPlan.enable_tile_exact = lambda self, on=True: False
    if args.merge_tail:
        import gslam_amd.plan as P11
        P11.TrackClosure.MERGE_TAIL = True
    if args.near_margin is not None:
        import gslam_amd.plan as P6
        P6.RenderPlan.NEAR_MARGIN = args.near_margin
    if args.no_near:
        import gslam_amd.plan as P7
        P7.RenderPlan.enable_near_placement = lambda self, margin=None: self.enable_defer_sort()
    if args.no_defer_sort:
        import gslam_amd.plan as P4
        P4.RenderPlan.enable_defer_sort = lambda self, margin=None: False
    if args.no_balance:
        import gslam_amd.plan as P0
        P0.RenderPlan.enable_balance = lambda self: False
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import BundleAdjuster, MapConfig
    from gslam_amd.plan import current_stream_ptr
    from gslam_amd.synthetic import make_scene
    from gslam_amd.tracking import GraphedTracker, TrackingConfig
    dev = torch.device("cuda:0")
    N, W, H = args.gaussians, 640, 480
    gt_scene = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
    m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    from gslam_amd.synthetic import sequence_param
    frames, cam = bench.make_frames(list(range(8)) + [sequence_param(i) for i in range(max(args.frames, 1))], W, H, dev, gt_scene)
    del gt_scene
    fm = m.no_grad_clone()
    conf = TrackingConfig()
    tr = GraphedTracker(fm, cam, conf)
    if args.front >= 0:
        tr.plan.r.front = bool(args.front)
    tr.track(frames[8])
    torch.cuda.synchronize()
    for i in range(args.frames):
        f = frames[8 + i]
        if args.eager:
            tr.load(f)
            tr.plan.init_optimizer(conf.n_adam_warmup, conf.pose_optim_lr, conf.lbfgs_history, 25)
            st = current_stream_ptr(dev)
            for _ in range(36):
                tr.plan.enqueue(st)
        else:
            tr.track(f, sync=False)
    torch.cuda.synchronize()
    assert tr.capacity_ok()
    r = tr.plan.r
    if getattr(r, "defer_sort", False):
        st4 = r.sort_stats.cpu().tolist()
        near, off = r.tile_near.cpu().numpy(), r.offsets.cpu().numpy()
        sizes = off[1:r.T + 1] - off[:r.T]
        if getattr(r, "row_keys", False):
            sizes = r.tile_span.cpu().numpy()[:, 1]
        if getattr(r, "near_place", False):
            print(f"near placement: {int(r.tile_placed.sum())} of {int(sizes.sum())} keys placed in the last closure, "
                  f"{st4[2]} tiles appended their far keys over all closures so far, {st4[3]} inconsistent")
        print(f"tile sort inside the rasteriser: {st4[0]} tile fall-backs, {st4[1]} through-memory sorts over all closures so far "
              f"({r.T} tiles per closure); last closure: {int(near.sum())} of {int(sizes.sum())} entries sorted, "
              f"largest near list {int(near.max())}, largest segment {int(sizes.max())}, margin {r.cut_margin}")
    if args.ba:
        ba = BundleAdjuster(m, MapConfig(), capturable=True)
        if args.ba_front >= 0:
            import gslam_amd.plan as P
            orig = P.RenderPlan.__init__

            def patched(self, *a, **k):
                k["front"] = bool(args.ba_front)
                orig(self, *a, **k)
            P.RenderPlan.__init__ = patched
        plan = ba.plan(frames[:args.ba_window])
        plan.prepare()
        for _ in range(args.ba):
            plan.step(graphed=not args.eager)
        torch.cuda.synchronize()
        assert plan.capacity_ok()
    print("done")


if __name__ == "__main__":
    main()
