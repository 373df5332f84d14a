"""End-to-end tracking + mapping rate on a synthetic TUM-shape sequence (north-star metric of BASELINE.json), with the
iteration counts of the reference: per tracked frame 10 Adam closures + one strong-Wolfe L-BFGS step (<= 25
evaluations) on the pose (gslam/frontend.py:613-658); every `kf_every` frames a keyframe is added and the backend runs
`ba_iters` bundle-adjustment iterations over the last <= 8 keyframes (gslam/backend.py:71-74).  One GPU runs both
(the reference also shares one device between its frontend and backend processes).

    python tools/bench_slam.py [--gaussians 500000] [--frames 20] [--kf-every 5] [--ba-iters 15]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--kf-every", type=int, default=5)
    ap.add_argument("--ba-iters", type=int, default=15)
    ap.add_argument("--window", type=int, default=8)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="run mapping and tracking one after the other on one stream (default: the BA iterations of a "
                         "keyframe run on their own HIP stream while the next frames are tracked, as the reference's "
                         "backend process does beside its frontend process)")
    ap.add_argument("--pose-lbfgs", action="store_true",
                    help="after the BA iterations of a keyframe also run the backend's window pose refinement "
                         "(backend.py:447-506: L-BFGS, <= 25 closures over the 8 keyframes) on the mapping stream")
    ap.add_argument("--host-optimizer", action="store_true",
                    help="keep torch.optim.Adam / LBFGS on the host (one loss.item() per closure, as the reference)")
    args = ap.parse_args()
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import BundleAdjuster, GraphedBundleAdjuster
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.rasterization import validate
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd.tracking import GraphedTracker, TrackingConfig

    dev = torch.device("cuda:0")
    W, H, N = 640, 480, args.gaussians
    K = make_intrinsics(W, H).to(dev)
    cam = Camera(K, H, W)
    gt_scene = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
    backend_map = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    frontend_map = backend_map.no_grad_clone()           # what sync() ships to the frontend (backend.py:508-519)

    def gt_frame(i):
        V = make_viewmat(i).to(dev)
        with torch.no_grad():
            img = gt_scene([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=False).rgbs[0].clamp(0, 1)
        return Frame(img=img.contiguous(), timestamp=i / 30.0, camera=cam, pose=PoseZhou(V).to(dev), gt_pose=V, index=i,
                     exposure_params=torch.zeros(2, device=dev))

    frames = [gt_frame(i) for i in range(args.frames + args.window)]
    keyframes = frames[:args.window]                       # pre-seeded window so that BA runs at its full size
    ba = BundleAdjuster(backend_map, capturable=not args.no_graph)
    tracker = GraphedTracker(frontend_map, cam, TrackingConfig(), device_optimizer=not (args.host_optimizer or args.no_graph))
    if args.no_graph:
        tracker.capture = lambda: None
    # warm-up (allocations, capacity probes, graph captures)
    tracker.track(frames[args.window])
    gba = None
    if not args.no_graph:
        gba = GraphedBundleAdjuster(ba, keyframes)
    else:
        ba.step(keyframes)
    refiner = None
    if args.pose_lbfgs and not args.no_graph:
        from gslam_amd.mapping import GraphedPoseRefiner
        refiner = GraphedPoseRefiner(backend_map, keyframes)
        refiner.run()                                       # capture
    torch.cuda.synchronize()
    assert validate(dev)

    n_closures, n_ba, n_refine = 0, 0, 0
    t_track = t_map = 0.0
    overlap = not (args.serial or args.no_graph)
    map_stream = torch.cuda.Stream() if overlap else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.frames):
        f = frames[args.window + i]
        ta = time.perf_counter()
        if overlap:
            rep = tracker.track(f, sync=False)              # no read-back: the frame is a fixed sequence of launches
            n_closures += tracker.conf.n_adam_warmup + tracker.max_eval + 1
        else:
            _, n = tracker.track(f)
            n_closures += n
            torch.cuda.synchronize()
        t_track += time.perf_counter() - ta
        if (i + 1) % args.kf_every == 0:
            tb = time.perf_counter()
            if overlap:
                # the backend's map is its own copy (backend.py:508-519 ships a clone to the frontend), so its BA
                # iterations only have to wait for the previous keyframe's; they overlap the tracking of later frames
                map_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(map_stream):
                    for _ in range(args.ba_iters):
                        gba.step()
                        n_ba += 1
                    if refiner is not None:
                        refiner.run_async()
                        n_refine += 1
            else:
                for _ in range(args.ba_iters):                 # same window object: images/poses updated in place
                    (gba.step() if gba is not None else ba.step(keyframes))
                    n_ba += 1
                torch.cuda.synchronize()
            t_map += time.perf_counter() - tb
    if overlap:
        torch.cuda.current_stream().wait_stream(map_stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ok = validate(dev)
    print(json.dumps({
        "metric": "tracking+mapping fps @640x480", "gaussians": N, "frames": args.frames, "fps": round(args.frames / elapsed, 2),
        "ms_per_frame": round(elapsed / args.frames * 1e3, 2), "closures_per_frame": round(n_closures / args.frames, 1),
        "ms_per_closure": None if overlap else round(t_track / max(n_closures, 1) * 1e3, 3), "ba_iters": n_ba,
        "pose_lbfgs_runs": n_refine, "ba_window": args.window, "ms_per_ba_iter": None if overlap else round(t_map / max(n_ba, 1) * 1e3, 3),
        "tracking_share": None if overlap else round(t_track / elapsed, 3),
        "launch": "eager" if args.no_graph else "hip-graph", "mapping": "own stream, overlapped" if overlap else "serial",
        "optimizer": "device state machine" if tracker.device_optimizer else "host torch.optim", "capacity_ok": ok}))


if __name__ == "__main__":
    main()
