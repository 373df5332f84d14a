"""SURVEY.md 8f rank 4 (host code, CPU): TUM sequence front on a synthetic sequence written to disk, Kabsch-Umeyama /
ATE, PSNR / SSIM as scikit-image defines them."""
import numpy as np
import pytest
import torch


def _write_sequence(root, n=4, name="rgbd_dataset_freiburg3_synthetic"):
    from PIL import Image
    d = root / name
    (d / "rgb").mkdir(parents=True)
    (d / "depth").mkdir()
    rng = np.random.default_rng(0)
    rgb_lines, depth_lines, gt_lines = ["# colour images"], ["# depth maps"], ["# ground truth"]
    imgs, deps = [], []
    for i in range(n):
        t = 100.0 + i / 30.0
        img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
        dep = rng.integers(2000, 20000, (480, 640), dtype=np.uint16)
        Image.fromarray(img).save(d / "rgb" / f"{t:.6f}.png")
        Image.fromarray(dep).save(d / "depth" / f"{t:.6f}.png")
        rgb_lines.append(f"{t:.6f} rgb/{t:.6f}.png")
        depth_lines.append(f"{t:.6f} depth/{t:.6f}.png")
        imgs.append(img); deps.append(dep)
    for j in range(3 * n):                                   # ground truth at 90 Hz: nearest-in-time association
        t = 100.0 + j / 90.0
        gt_lines.append(f"{t:.6f} {0.01 * j:.4f} {0.02 * j:.4f} 0.5 0.0 0.0 {np.sin(0.01 * j):.6f} {np.cos(0.01 * j):.6f}")
    (d / "rgb.txt").write_text("\n".join(rgb_lines) + "\n")
    (d / "depth.txt").write_text("\n".join(depth_lines) + "\n")
    (d / "groundtruth.txt").write_text("\n".join(gt_lines) + "\n")
    return d, imgs, deps


def test_tum_sequence_front(tmp_path):
    from gslam_amd.data import TumRGB
    d, imgs, deps = _write_sequence(tmp_path)
    seq = TumRGB(d, device="cpu")
    assert len(seq) == 4 and TumRGB(d, seq_len=2, device="cpu").length == 2
    assert seq.roi == (0, 0, 639, 479)                       # OpenCV's valid-pixel rectangle for a distortion-free lens
    np.testing.assert_allclose(seq.Ks.numpy(), [[535.4, 0, 320.1], [0, 539.2, 247.6], [0, 0, 1]], rtol=1e-6)
    f = seq[2]
    assert f.img.shape == (479, 639, 3) and f.gt_depth.shape == (479, 639) and f.camera.width == 639
    np.testing.assert_array_equal((f.img.numpy() * 255).round().astype(np.uint8), imgs[2][:479, :639])
    np.testing.assert_allclose(f.gt_depth.numpy(), deps[2][:479, :639].astype(np.float32) / 5000.0)
    j = 6                                                    # frame 2 at t0 + 2/30 = ground-truth sample 6 at t0 + 6/90
    np.testing.assert_allclose(f.gt_pose[:3, 3].numpy(), [0.01 * j, 0.02 * j, 0.5], atol=1e-6)
    c, s = np.cos(0.02 * j), np.sin(0.02 * j)                # quaternion (0,0,sin a,cos a) = rotation by 2a about z
    np.testing.assert_allclose(f.gt_pose[:3, :3].numpy(), [[c, -s, 0], [s, c, 0], [0, 0, 1]], atol=1e-5)
    with pytest.raises(StopIteration):
        seq[4]


def test_undistortion_restatement_is_self_consistent():
    from gslam_amd.data import (_distort, _undistort_points, optimal_new_camera_matrix, remap_bilinear,
                                tum_intrinsics_params, undistort_maps)
    fx, fy, cx, cy, *d = tum_intrinsics_params["freiburg1"]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    newK, roi = optimal_new_camera_matrix(K, d, (640, 480))
    x, y, w, h = roi
    assert 0 <= x < 40 and 0 <= y < 40 and 560 < w <= 640 and 400 < h <= 480          # alpha = 0: a mild crop / zoom
    mx, my = undistort_maps(K, d, newK, (640, 480))
    assert mx.min() >= -1.0 and mx.max() <= 640.0 and my.min() >= -1.0 and my.max() <= 480.0   # every pixel valid
    # distort(undistort(p)) = p for points near the centre (where the 5-step fixed-point iteration has converged)
    px, py = np.meshgrid(np.linspace(200, 440, 7), np.linspace(150, 330, 7))
    ux, uy = _undistort_points(px, py, K, d)
    bx, by = _distort(ux, uy, d)
    np.testing.assert_allclose(bx * fx + cx, px, atol=5e-2)
    np.testing.assert_allclose(by * fy + cy, py, atol=5e-2)
    img = np.random.default_rng(1).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    ident_x, ident_y = np.meshgrid(np.arange(640, dtype=np.float32), np.arange(480, dtype=np.float32))
    np.testing.assert_array_equal(remap_bilinear(img, ident_x, ident_y), img)
    half = remap_bilinear(img, ident_x + 0.5, ident_y)
    expect = np.rint((img[:, :-1].astype(np.float64) + img[:, 1:]) / 2)
    np.testing.assert_array_equal(half[:, :-1], expect.astype(np.uint8))


def test_kabsch_umeyama_and_ate():
    from gslam_amd.trajectory import align, average_translation_error, evaluate_trajectories, kabsch_umeyama
    rng = np.random.default_rng(3)
    B = rng.normal(size=(50, 3))
    a = 0.7
    Rt = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    A = np.array([0.3, -1.0, 2.0]) + 2.5 * (B @ Rt.T)
    R, c, t = kabsch_umeyama(A, B)
    np.testing.assert_allclose(R, Rt, atol=1e-9)
    assert abs(c - 2.5) < 1e-9
    np.testing.assert_allclose(t, [0.3, -1.0, 2.0], atol=1e-9)
    assert average_translation_error(A, B) < 1e-9
    noisy = B + 0.01 * rng.normal(size=B.shape)
    e = average_translation_error(A, noisy)
    assert 0.01 < e < 0.08                                   # 2.5 x the noise's mean norm (~0.016 per axis-triple)
    np.testing.assert_allclose(align(A, B), A, atol=1e-9)

    class F:
        def __init__(self, gt, est):
            self.gt_pose = torch.eye(4); self.gt_pose[:3, 3] = torch.tensor(gt, dtype=torch.float32)
            self._e = torch.eye(4); self._e[:3, 3] = torch.tensor(est, dtype=torch.float32)

        def pose(self):
            return self._e
    frames = [F(A[i], B[i]) for i in range(20)]
    out = evaluate_trajectories({"tracking": frames, "short": frames[:1]})
    assert set(out) == {"ate_tracking"} and out["ate_tracking"] < 1e-5


def test_psnr_ssim_match_their_definitions():
    from gslam_amd.evaluation import psnr_uint8, ssim_uint8, to_uint8
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 256, (40, 48, 3), dtype=np.uint8)
    img = np.clip(ref.astype(np.int32) + rng.integers(-20, 21, ref.shape), 0, 255).astype(np.uint8)
    mse = np.mean((img.astype(np.float64) - ref) ** 2)
    assert abs(psnr_uint8(img, ref) - 10 * np.log10(255 ** 2 / mse)) < 1e-12 and psnr_uint8(ref, ref) == float('inf')
    assert abs(ssim_uint8(ref, ref) - 1.0) < 1e-12
    # brute force: 7x7 windows fully inside the image, sample covariance, per channel, mean
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    acc = []
    for ch in range(3):
        x, y = img[..., ch].astype(np.float64), ref[..., ch].astype(np.float64)
        for i in range(3, 40 - 3):
            for j in range(3, 48 - 3):
                wx, wy = x[i - 3:i + 4, j - 3:j + 4].ravel(), y[i - 3:i + 4, j - 3:j + 4].ravel()
                ux, uy = wx.mean(), wy.mean()
                vx, vy = wx.var(ddof=1), wy.var(ddof=1)
                vxy = ((wx - ux) * (wy - uy)).sum() / 48.0
                acc.append(((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2)))
    assert abs(ssim_uint8(img, ref) - np.mean(acc)) < 1e-9
    assert to_uint8(torch.tensor([[[0.5, 1.2, -0.1]]])).tolist() == [[[127, 255, 0]]]


def test_trajectory_evaluation_matches_reference_fixture():
    """gslam_amd.trajectory against tests/golden/trajectory.npz, written by oracle/gen_golden.py from the IMPORTED reference
    (gslam/trajectory.py:14-97: kabsch_umeyama, average_translation_error, evaluate_trajectories): a similarity-transformed
    noisy trajectory, a mirrored one (the determinant correction of the SVD) and a three-point one."""
    import os
    import numpy as np
    import torch
    from gslam_amd.trajectory import average_translation_error, evaluate_trajectories, kabsch_umeyama
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "trajectory.npz"))
    for name in ("similar", "mirrored", "short"):
        A, B = g[f"{name}__A"], g[f"{name}__B"]
        R, c, t = kabsch_umeyama(A, B)
        np.testing.assert_allclose(R, g[f"{name}__R"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(c, float(g[f"{name}__c"]), rtol=1e-12)
        np.testing.assert_allclose(t, g[f"{name}__t"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(average_translation_error(A, B), float(g[f"{name}__ate"]), rtol=1e-10, atol=1e-14)

    class F:
        def __init__(self, gt, est):
            self.gt_pose, self._est = torch.from_numpy(gt), torch.from_numpy(est)

        def pose(self):
            return self._est

    def frames(a, b):
        out = []
        for i in range(len(a)):
            gt, e = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
            gt[:3, 3], e[:3, 3] = a[i], b[i]
            out.append(F(gt, e))
        return out
    A, B = g["similar__A"], g["similar__B"]
    ates = evaluate_trajectories({"tracking": frames(A, B), "keyframes": frames(A[::5], B[::5])}, keyframe_indices=[0, 5, 10])
    assert set(ates) == {"ate_tracking", "ate_keyframes"}
    for k, v in ates.items():
        np.testing.assert_allclose(v, float(g[f"eval__{k}"]), rtol=1e-6)
