"""The frontend / backend pair behind the reference's message API (gslam/frontend.py, gslam/backend.py) on a synthetic
TUM-shape sequence: REQUEST_INIT -> SYNC -> tracked frames (ADD_FRAME) -> keyframes, BA, pruning, SYNC -> None ->
END_SYNC.  Driven in lockstep in one thread through Backend.handle / idle_step (the same code Backend.run loops over).
Run with -m gpu."""
import queue

import pytest
import torch

pytestmark = pytest.mark.gpu

# bounds of the end-to-end assertion below: ~3x the values measured on MI355X (printed by the test)
ATE_BOUND_M, PSNR_BOUND_DB, SSIM_BOUND = 0.04, 17.0, 0.6      # measured: 1.2 cm, 21.5 dB, 0.77 (13 frames, 0.3 m path)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_slam_loop_messages_and_tracking(dev):
    from gslam_amd.backend import Backend, MapConfig
    from gslam_amd.frontend import Frontend
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.messages import BackendMessage, FrontendMessage
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd.tracking import TrackingConfig
    torch.manual_seed(0)
    W, H = 320, 240
    K = make_intrinsics(W, H).to(dev)
    cam = Camera(K, H, W)
    sc = make_scene(30000, 3)
    sc["scales"] = sc["scales"] + 0.6
    world = GaussianSplattingData.from_dict(sc, dev)

    def sensor_frame(i):
        V = make_viewmat(i).to(dev)
        V[:3, 3] *= 0.5                                    # gentle motion: 2.5 cm and 1 degree per frame
        with torch.no_grad():
            img = world([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=False).rgbs[0].clamp(0, 1)
        return Frame(img=img.contiguous(), timestamp=i / 30.0, camera=cam, pose=None, gt_pose=V, index=i)

    to_backend, to_frontend, sensor = queue.Queue(), queue.Queue(), queue.Queue()
    conf = MapConfig(num_iters_initialization=60, num_iters_mapping=5, kf_m=0.02)
    be = Backend(conf, to_backend, to_frontend)
    fe = Frontend(TrackingConfig(), to_backend, to_frontend, sensor)

    def pump_backend():
        while not to_backend.empty():
            assert be.handle(to_backend.get())
        while not to_frontend.empty():
            fe.handle_message_from_backend(to_frontend.get())

    # frame 0: REQUEST_INIT -> the backend initialises a 5000-splat map from the mock depth map and SYNCs
    fe.track(sensor_frame(0).to(dev))
    assert fe.waiting_for_sync and to_backend.qsize() == 1
    msg = to_backend.queue[0]
    assert msg[0] == FrontendMessage.REQUEST_INIT and str(msg[0]) == "request_init"
    pump_backend()
    assert not fe.waiting_for_sync and fe.splats is not None and 0 in fe.keyframes
    n_init = be.splats.means.shape[0]
    assert 2000 < n_init <= 5000 and be.splats.ages.dtype == torch.int64
    assert fe.splats.means.shape[0] == n_init and not fe.splats.means.requires_grad

    # frames 1..12: tracked against the frontend's copy, shipped with ADD_FRAME; the backend adds keyframes, maps, syncs
    n_sync = 0
    for i in range(1, 13):
        fe.track(sensor_frame(i).to(dev))
        assert to_backend.queue[-1][0] == FrontendMessage.ADD_FRAME
        before = to_frontend.qsize()
        while not to_backend.empty():
            assert be.handle(to_backend.get())
        n_sync += to_frontend.qsize() - before
        be.idle_step()
        while not to_frontend.empty():
            m = to_frontend.get()
            assert m[0] == BackendMessage.SYNC and len(m) == 6
            fe.handle_message_from_backend(m)
    torch.cuda.synchronize()
    assert n_sync >= 2                                      # frame.index % 5 == 0 (backend.py:864)
    assert len(be.keyframes) >= 3 and len(fe.frames) == 13
    assert all(k in be.pose_graph for k in be.keyframes)
    assert all(torch.isfinite(torch.tensor(l)) for l in fe.last_losses)
    # the map was optimised and pruned in place: still a consistent set of per-Gaussian arrays + Adam state
    n = be.splats.means.shape[0]
    for name, p in be.splats.named_parameters():
        assert p.shape[0] == n, name
    st = be.ba.optimizers.splat_opt.state[be.splats.means]
    assert st["exp_avg"].shape[0] == n
    # ---- end-to-end quality on the synthetic sequence (gslam/trajectory.py:14-97, gslam/frontend.py:374-409) ---------------------
    # the map starts from a MOCK depth map (monocular initialisation, backend.py:604-660), so scale is free: the absolute
    # trajectory error is taken after the similarity alignment the reference uses (Kabsch-Umeyama).  The camera moves 2.5 cm
    # and 1 degree per frame over 13 frames (path length 0.3 m).
    import numpy as np
    from gslam_amd.evaluation import evaluate_reconstruction, to_uint8
    from gslam_amd.trajectory import evaluate_trajectories
    ates = evaluate_trajectories({"frontend": fe.frames, "keyframes": list(be.keyframes.values())})
    kfs = list(be.keyframes.values())
    rec = evaluate_reconstruction(be.splats, kfs, [to_uint8(f.img) for f in kfs])
    print(f"slam loop quality: ATE frontend {ates['ate_frontend'] * 100:.2f} cm, keyframes {ates['ate_keyframes'] * 100:.2f} cm "
          f"over {len(fe.frames)} frames; keyframe renders PSNR {rec['psnr']:.2f} dB, SSIM {rec['ssim']:.3f} "
          f"({n} Gaussians after {be.total_step} mapping iterations)")
    assert np.isfinite(ates["ate_frontend"]) and ates["ate_frontend"] < ATE_BOUND_M
    assert rec["psnr"] > PSNR_BOUND_DB and rec["ssim"] > SSIM_BOUND
    # end of stream: None -> END_SYNC with the final map
    to_backend.put(None)
    assert not be.handle(to_backend.get())
    be.end_sync()
    m = to_frontend.get()
    assert m[0] == BackendMessage.END_SYNC and len(m) == 3
    fe.handle_message_from_backend(m)
    assert fe.done and fe.splats.means.shape[0] == n


def test_map_mailbox_double_buffer_and_in_place_receive(dev):
    """the SYNC payload's map as a view of a double-buffered device slot (one launch per publish) and
    an in-place, one-launch receive that keeps the consumer's tensor addresses while N is unchanged"""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.synthetic import make_scene
    from gslam_amd.transport import MapMailbox, receive
    src = GaussianSplattingData.from_dict(make_scene(5000, 0), dev)
    box = MapMailbox()
    a = box.publish(src)
    for p in GaussianSplattingData._per_splat_params:
        assert torch.equal(getattr(a, p), getattr(src, p)) and not getattr(a, p).requires_grad
    assert a.ages.dtype == torch.int64
    mine, replaced = receive(None, a)
    assert replaced and torch.equal(mine.means, src.means) and mine.means.data_ptr() != a.means.data_ptr()
    ptrs = [getattr(mine, p).data_ptr() for p in GaussianSplattingData._per_splat_params]
    with torch.no_grad():
        src.means.add_(1.0)
        src.ages.add_(3)
    b = box.publish(src)                                    # second slot: the first payload is still intact
    assert b.means.data_ptr() != a.means.data_ptr()
    assert torch.equal(a.means + 1.0, b.means) and torch.equal(a.ages + 3, b.ages)
    mine2, replaced = receive(mine, b)
    assert mine2 is mine and not replaced
    assert [getattr(mine, p).data_ptr() for p in GaussianSplattingData._per_splat_params] == ptrs
    assert torch.equal(mine.means, src.means) and torch.equal(mine.ages, src.ages)
    c = box.publish(src)                                    # third publish reuses the first slot
    assert c.means.data_ptr() == a.means.data_ptr()
    bigger = GaussianSplattingData.from_dict(make_scene(7000, 1), dev)
    d = box.publish(bigger)                                 # grown map: the slot is re-allocated, the receiver replaces
    mine3, replaced = receive(mine, d)
    assert replaced and mine3.means.shape[0] == 7000 and torch.equal(mine3.quats, bigger.quats)


def test_slam_loop_from_a_tum_directory(dev, tmp_path):
    """SURVEY 8f rank 4 consumed on the GPU: a synthetic sequence WRITTEN in the TUM RGB-D layout (rgb / depth PNGs, rgb.txt,
    depth.txt, groundtruth.txt with camera-to-world poses as tx ty tz qx qy qz qw at 90 Hz) -> gslam_amd.data.TumRGB (nearest
    pose in time, freiburg3 intrinsics, valid-pixel rectangle 639x479: image sizes that are not multiples of the tile) ->
    Frontend / Backend over the message API -> trajectory against the file's ground truth (camera centres, similarity-aligned:
    the map starts from a mock depth, scale is free)."""
    import numpy as np
    from PIL import Image
    from scipy.spatial.transform import Rotation
    from gslam_amd.backend import Backend, MapConfig
    from gslam_amd.data import TumRGB, tum_intrinsics_params
    from gslam_amd.frontend import Frontend
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.primitives import Camera, PoseZhou
    from gslam_amd.synthetic import make_scene, make_viewmat
    from gslam_amd.tracking import TrackingConfig
    from gslam_amd.trajectory import average_translation_error
    torch.manual_seed(0)
    n_frames, W, H = 9, 640, 480
    fx, fy, cx, cy, *dist = tum_intrinsics_params["freiburg3"]
    assert not any(dist)
    K = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32, device=dev)
    cam = Camera(K, H, W)
    sc = make_scene(30000, 3)
    sc["scales"] = sc["scales"] + 0.6
    world = GaussianSplattingData.from_dict(sc, dev)
    d = tmp_path / "rgbd_dataset_freiburg3_synthetic"
    (d / "rgb").mkdir(parents=True)
    (d / "depth").mkdir()
    rgb_lines, depth_lines, gt_lines, centres = ["# colour"], ["# depth"], ["# ground truth"], []
    for i in range(n_frames):
        V = make_viewmat(i).to(dev)
        V[:3, 3] *= 0.5
        with torch.no_grad():
            out = world([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=True)
        img = (out.rgbs[0].clamp(0, 1) * 255.0).round().to(torch.uint8).cpu().numpy()
        dep = (out.depthmaps[0].clamp(0, 13.0) * 5000.0).round().to(torch.int32).cpu().numpy().astype(np.uint16)
        t = 200.0 + i / 30.0
        Image.fromarray(img).save(d / "rgb" / f"{t:.6f}.png")
        Image.fromarray(dep).save(d / "depth" / f"{t:.6f}.png")
        rgb_lines.append(f"{t:.6f} rgb/{t:.6f}.png")
        depth_lines.append(f"{t:.6f} depth/{t:.6f}.png")
        c2w = torch.linalg.inv(V.cpu().double()).numpy()
        q = Rotation.from_matrix(c2w[:3, :3]).as_quat()                       # x y z w, the TUM order
        centres.append(c2w[:3, 3])
        for dt in (-1.0 / 90.0, 0.0, 1.0 / 90.0):                             # 90 Hz ground truth; the reader takes the nearest
            p = c2w[:3, 3] + (0.0 if dt == 0.0 else 0.5)                       # (the off-time samples are decoys)
            gt_lines.append(f"{t + dt:.6f} {p[0]:.6f} {p[1]:.6f} {p[2]:.6f} {q[0]:.8f} {q[1]:.8f} {q[2]:.8f} {q[3]:.8f}")
    (d / "rgb.txt").write_text("\n".join(rgb_lines) + "\n")
    (d / "depth.txt").write_text("\n".join(depth_lines) + "\n")
    (d / "groundtruth.txt").write_text("\n".join(gt_lines) + "\n")

    seq = TumRGB(d, device=dev)
    assert len(seq) == n_frames and seq.roi == (0, 0, 639, 479)
    to_backend, to_frontend, sensor = queue.Queue(), queue.Queue(), queue.Queue()
    be = Backend(MapConfig(num_iters_initialization=60, num_iters_mapping=5, kf_m=0.02), to_backend, to_frontend)
    fe = Frontend(TrackingConfig(), to_backend, to_frontend, sensor)
    for i in range(n_frames):
        f = seq[i]
        assert tuple(f.img.shape) == (479, 639, 3) and f.img.is_cuda
        np.testing.assert_allclose(f.gt_pose[:3, 3].cpu().numpy(), centres[i], atol=1e-5)
        fe.track(f)
        while not to_backend.empty():
            assert be.handle(to_backend.get())
        if i:
            be.idle_step()
        while not to_frontend.empty():
            fe.handle_message_from_backend(to_frontend.get())
    torch.cuda.synchronize()
    assert len(fe.frames) == n_frames and len(be.keyframes) >= 2
    est = np.array([np.linalg.inv(fr.pose().detach().cpu().double().numpy())[:3, 3] for fr in fe.frames])
    gt = np.array([fr.gt_pose[:3, 3].cpu().double().numpy() for fr in fe.frames])
    ate = average_translation_error(gt, est)
    path = float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum())
    print(f"TUM-directory loop: {n_frames} frames of 639x479, path {path * 100:.1f} cm, ATE (camera centres) {ate * 100:.2f} cm, "
          f"{be.splats.means.shape[0]} Gaussians, {len(be.keyframes)} keyframes")
    assert np.isfinite(ate) and ate < ATE_BOUND_M
