"""The reference's process topology (main.py:61-91): Frontend and Backend as two SPAWNED processes on one GPU, talking through
torch.multiprocessing queues with the reference's message tuples.  Tensors in the messages - frames going one way, the map,
the keyframes, the last keyframe's depth map and render coming back - cross the process boundary as device memory (HIP IPC),
never through the host.  Run with -m gpu."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _backend_proc(to_backend, to_frontend, done, res):
    import torch
    torch.cuda.set_device(0)
    from gslam_amd.backend import Backend, MapConfig
    try:
        torch.manual_seed(0)
        conf = MapConfig(num_iters_initialization=60, num_iters_mapping=5, kf_m=0.02)
        be = Backend(conf, to_backend, to_frontend, backend_done_event=done)
        be.run()                                            # the reference's loop: messages, idle mapping cycles, END_SYNC
        torch.cuda.synchronize()
        res.put(("backend", dict(keyframes=len(be.keyframes), n=int(be.splats.means.shape[0]), steps=int(be.total_step))))
    except BaseException as e:  # noqa: BLE001
        res.put(("backend", dict(error=repr(e))))
        done.set()
        raise


def _frontend_proc(to_backend, to_frontend, done, res, n_frames):
    import queue

    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from gslam_amd.frontend import Frontend
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.messages import BackendMessage
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd.tracking import TrackingConfig
    try:
        W, H = 320, 240
        cam = Camera(make_intrinsics(W, H).to(dev), H, W)
        sc = make_scene(30000, 3)
        sc["scales"] = sc["scales"] + 0.6
        world = GaussianSplattingData.from_dict(sc, dev)
        sensor = queue.Queue()
        for i in range(n_frames):
            V = make_viewmat(i).to(dev)
            V[:3, 3] *= 0.5
            with torch.no_grad():
                img = world([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=False).rgbs[0].clamp(0, 1)
            sensor.put(Frame(img=img.contiguous(), timestamp=i / 30.0, camera=cam, pose=None, gt_pose=V, index=i))
        sensor.put(None)
        fe = Frontend(TrackingConfig(), to_backend, to_frontend, sensor, backend_done_event=done)
        seen = dict(sync=0, map_on_device=True, depth_on_device=True, kf_on_device=True, end_sync=0)
        handle = fe.handle_message_from_backend

        def spy(message):
            if message[0] == BackendMessage.SYNC:
                seen["sync"] += 1
                seen["map_on_device"] &= bool(message[4].means.is_cuda and message[4].ages.is_cuda)
                seen["depth_on_device"] &= bool(message[2].is_cuda and message[3].is_cuda)
                seen["kf_on_device"] &= all(f.img.is_cuda and f.pose.Rt.is_cuda for f in message[1].values())
            elif message[0] == BackendMessage.END_SYNC:
                seen["end_sync"] += 1
                seen["map_on_device"] &= bool(message[1].means.is_cuda)
            return handle(message)

        fe.handle_message_from_backend = spy
        fe.run(timeout_s=240.0)
        torch.cuda.synchronize()
        from gslam_amd.trajectory import evaluate_trajectories
        ate = evaluate_trajectories({"frontend": fe.frames})["ate_frontend"]
        res.put(("frontend", dict(seen, frames=len(fe.frames), done=bool(fe.done), n=int(fe.splats.means.shape[0]),
                                   finite=all(l == l and abs(l) < 1e30 for l in fe.last_losses), ate=float(ate))))
    except BaseException as e:  # noqa: BLE001
        res.put(("frontend", dict(error=repr(e))))
        raise


def test_frontend_and_backend_as_two_processes():
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (the only mode this driver supports)
    ctx = mp.get_context("spawn")
    to_backend, to_frontend, res = ctx.Queue(), ctx.Queue(), ctx.Queue()
    done = ctx.Event()
    n_frames = 13
    pb = ctx.Process(target=_backend_proc, args=(to_backend, to_frontend, done, res))
    pf = ctx.Process(target=_frontend_proc, args=(to_backend, to_frontend, done, res, n_frames))
    pb.start()
    pf.start()
    out = dict(res.get(timeout=420) for _ in range(2))
    for p in (pf, pb):
        p.join(timeout=120)
    assert "error" not in out["frontend"], out["frontend"]
    assert "error" not in out["backend"], out["backend"]
    assert pf.exitcode == 0 and pb.exitcode == 0
    fe, be = out["frontend"], out["backend"]
    assert fe["frames"] == n_frames and fe["done"] and fe["end_sync"] == 1          # END_SYNC received, loop ended
    assert fe["sync"] >= 2                                                           # initialisation + frame.index % 5 == 0
    assert fe["map_on_device"] and fe["depth_on_device"] and fe["kf_on_device"]      # nothing came through the host
    assert fe["finite"] and fe["n"] == be["n"] > 1000 and be["keyframes"] >= 2 and be["steps"] >= 60
    print(f"two processes: {fe['frames']} frames, {fe['sync']} SYNCs, {be['keyframes']} keyframes, {be['n']} Gaussians, "
          f"{be['steps']} mapping iterations, ATE {fe['ate'] * 100:.2f} cm")
    assert fe["ate"] < 0.05
