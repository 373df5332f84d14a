"""The whole loop on a synthetic TUM-shape sequence, wired like the reference's main.py:38-95 but in ONE process (one
process per GPU): the Frontend runs in the main thread on the default HIP stream, the Backend in a second thread on
its own stream, talking through the reference's message tuples over queue.Queue.

    python tools/run_slam.py [--frames 60] [--world 30000] [--width 640 --height 480]
"""
import argparse
import json
import os
import queue
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--world", type=int, default=30000, help="Gaussians of the synthetic world the sensor renders")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--init-iters", type=int, default=400)
    args = ap.parse_args()
    from gslam_amd.backend import Backend, MapConfig
    from gslam_amd.frontend import Frontend
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd.tracking import TrackingConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    W, H = args.width, args.height
    K = make_intrinsics(W, H).to(dev)
    cam = Camera(K, H, W)
    sc = make_scene(args.world, 3)
    sc["scales"] = sc["scales"] + 0.6
    world = GaussianSplattingData.from_dict(sc, dev)
    frames = []
    for i in range(args.frames):
        V = make_viewmat(i).to(dev)
        V[:3, 3] *= 0.5
        with torch.no_grad():
            img = world([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=False).rgbs[0].clamp(0, 1)
        frames.append(Frame(img=img.contiguous(), timestamp=i / 30.0, camera=cam, pose=None, gt_pose=V, index=i))
    torch.cuda.synchronize()

    to_backend, to_frontend, sensor = queue.Queue(), queue.Queue(), queue.Queue()
    be_done = threading.Event()
    conf = MapConfig(num_iters_initialization=args.init_iters)
    be = Backend(conf, to_backend, to_frontend, backend_done_event=be_done)
    fe = Frontend(TrackingConfig(), to_backend, to_frontend, sensor, backend_done_event=be_done)
    be_stream = torch.cuda.Stream()
    errors = []

    def backend_thread():
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(be_stream):
                be.run()
        except BaseException as e:      # noqa: BLE001  (report and unblock the frontend)
            errors.append(repr(e))
            be_done.set()
            to_frontend.put(("end_sync", be.splats, be.keyframes))

    th = threading.Thread(target=backend_thread, daemon=True)
    th.start()
    for f in frames:
        sensor.put(f)
    sensor.put(None)
    t0 = time.perf_counter()
    fe.run(timeout_s=120.0)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    th.join(timeout=30.0)
    n_tracked = len(fe.frames)
    print(json.dumps({
        "metric": "frontend + backend loop fps (reference message API, synthetic sequence)", "frames": n_tracked,
        "fps": round(n_tracked / elapsed, 2), "seconds": round(elapsed, 2), "keyframes": len(be.keyframes),
        "map_gaussians": int(be.splats.means.shape[0]), "ba_steps": be.total_step, "width": W, "height": H,
        "errors": errors}))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
