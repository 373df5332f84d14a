"""TUM RGB-D sequence front of gslam/data.py:67-207: `rgb.txt` / `depth.txt` / `groundtruth.txt`
association, lens undistortion + valid-pixel crop, depth PNG / 5000, Frame construction.

The reference leans on OpenCV (`getOptimalNewCameraMatrix(alpha=0)`, `initUndistortRectifyMap`, `remap`) and
pyquaternion; neither is a dependency here, so those steps are restated from their documented algorithms in numpy
(PARITY UNPINNED for the distorted freiburg1 / freiburg2 sequences: no OpenCV in this image to compare against; for
freiburg3 - zero distortion, `data.py:36` - the maps are the identity and the crop is OpenCV's (0, 0, W-1, H-1)).
Not on the hot path: host code, runs once per frame."""
from __future__ import annotations

from pathlib import Path
from typing import Tuple

import numpy as np
import torch

from .primitives import Camera, Frame, PoseZhou

# fx, fy, cx, cy, k1, k2, p1, p2, k3 (gslam/data.py:23-37; the TUM calibration page's values)
tum_intrinsics_params = {
    "freiburg1": [517.3, 516.5, 318.6, 255.3, 0.2624, -0.9531, -0.0054, 0.0026, 1.1633],
    "freiburg2": [520.9, 521.0, 325.1, 249.7, 0.2312, -0.7849, -0.0033, -0.0001, 0.9172],
    "freiburg3": [535.4, 539.2, 320.1, 247.6, 0, 0, 0, 0, 0],
}


def quat_xyzw_to_matrix(q: np.ndarray) -> np.ndarray:
    """rotation matrix of a (normalised) quaternion given as x, y, z, w (TUM order, data.py:101-106)"""
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def _distort(x, y, d):
    k1, k2, p1, p2, k3 = d
    r2 = x * x + y * y
    radial = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    return (x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y)


def _undistort_points(px, py, K, d, newK=None, iters: int = 5):
    """pixel -> ideal coordinates (normalised, or pixels of newK): OpenCV undistortPoints' fixed-point iteration"""
    k1, k2, p1, p2, k3 = d
    x0, y0 = (px - K[0, 2]) / K[0, 0], (py - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx, dy = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    if newK is not None:
        x, y = x * newK[0, 0] + newK[0, 2], y * newK[1, 1] + newK[1, 2]
    return x, y


def _inner_rect(K, d, size: Tuple[int, int], newK=None):
    """largest axis-aligned rectangle inside the undistorted image border, from a 9x9 grid of border samples"""
    W, H = size
    n = 9
    gx, gy = np.meshgrid(np.arange(n) * (W - 1) / (n - 1), np.arange(n) * (H - 1) / (n - 1))
    ux, uy = _undistort_points(gx.astype(np.float64), gy.astype(np.float64), K, d, newK)
    x0, x1 = ux[:, 0].max(), ux[:, n - 1].min()
    y0, y1 = uy[0, :].max(), uy[n - 1, :].min()
    return x0, y0, x1 - x0, y1 - y0


def optimal_new_camera_matrix(K: np.ndarray, d, size: Tuple[int, int]):
    """cv2.getOptimalNewCameraMatrix(K, d, size, alpha=0, newImgSize=size) -> (newK [3,3] float32-ish, roi (x, y, w, h))"""
    W, H = size
    ix, iy, iw, ih = _inner_rect(K, d, size)
    fx, fy = (W - 1) / iw, (H - 1) / ih
    newK = np.array([[fx, 0, -fx * ix], [0, fy, -fy * iy], [0, 0, 1]], dtype=np.float64)
    rx, ry, rw, rh = _inner_rect(K, d, size, newK)
    x, y, w, h = int(np.rint(rx)), int(np.rint(ry)), int(np.rint(rw)), int(np.rint(rh))
    x2, y2 = min(x + w, W), min(y + h, H)
    x, y = max(x, 0), max(y, 0)
    return newK, (x, y, max(x2 - x, 0), max(y2 - y, 0))


def undistort_maps(K: np.ndarray, d, newK: np.ndarray, size: Tuple[int, int]):
    """cv2.initUndistortRectifyMap(K, d, None, newK, size, CV_32FC1): source pixel of every destination pixel"""
    W, H = size
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x, y = (u - newK[0, 2]) / newK[0, 0], (v - newK[1, 2]) / newK[1, 1]
    xd, yd = _distort(x, y, d)
    return (xd * K[0, 0] + K[0, 2]).astype(np.float32), (yd * K[1, 1] + K[1, 2]).astype(np.float32)


def remap_bilinear(img: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(img, map_x, map_y, INTER_LINEAR) with the default constant (0) border; uint8 [H,W,C] in and out
    (float interpolation, round to nearest: OpenCV's 5-bit fixed-point weights can differ by one grey level)"""
    H, W = img.shape[:2]
    x0, y0 = np.floor(map_x).astype(np.int64), np.floor(map_y).astype(np.int64)
    fx, fy = (map_x - x0)[..., None].astype(np.float64), (map_y - y0)[..., None].astype(np.float64)
    src = img.astype(np.float64)

    def at(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        out = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return out * ok[..., None]

    val = (at(y0, x0) * (1 - fx) * (1 - fy) + at(y0, x0 + 1) * fx * (1 - fy) + at(y0 + 1, x0) * (1 - fx) * fy +
           at(y0 + 1, x0 + 1) * fx * fy)
    return np.clip(np.rint(val), 0, 255).astype(np.uint8)


class TumRGB:
    """`TumRGB(sequence_dir, seq_len)`; `len()`, `[i]` -> Frame with img [h,w,3] float in [0,1], gt_depth [h,w] metres,
    gt_pose [4,4] (nearest ground-truth pose in time, camera to world as in the TUM files), camera with the new K."""

    def __init__(self, sequence_dir, seq_len: int = -1, device="cuda"):
        from PIL import Image
        self._Image = Image
        self.sequence_dir = Path(sequence_dir)
        self.device = torch.device(device)
        rgb = np.loadtxt(self.sequence_dir / "rgb.txt", np.str_, ndmin=2)
        self.rgb_frame_timestamps = rgb[:, 0].astype(np.float64)
        self.rgb_frame_filenames = rgb[:, 1]
        depth = np.loadtxt(self.sequence_dir / "depth.txt", np.str_, ndmin=2)
        self.depth_frame_timestamps = depth[:, 0].astype(np.float64)
        self.depth_frame_filenames = depth[:, 1]
        self.num_frames = len(self.rgb_frame_filenames)
        gt = np.loadtxt(self.sequence_dir / "groundtruth.txt", np.str_, ndmin=2)
        gt_t, gt_p = gt[:, 0].astype(np.float64), gt[:, 1:].astype(np.float64)
        nearest = np.abs(np.subtract.outer(self.rgb_frame_timestamps, gt_t)).argmin(axis=1)      # data.py:89-97
        self.poses = np.tile(np.eye(4, dtype=np.float64), [self.num_frames, 1, 1])
        self.poses[:, :3, :3] = np.array([quat_xyzw_to_matrix(q) for q in gt_p[nearest][:, 3:]])
        self.poses[:, :3, 3] = gt_p[nearest][:, :3]
        self.length = self.num_frames if seq_len <= 0 else min(self.num_frames, seq_len)
        sequence_type = str(self.sequence_dir.parts[-1]).split('_')[2]                             # data.py:112
        fx, fy, cx, cy, *d = tum_intrinsics_params[sequence_type]
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
        self.distortion = d
        newK, self.roi = optimal_new_camera_matrix(K, d, (640, 480))
        self.undistort_map_x, self.undistort_map_y = undistort_maps(K, d, newK, (640, 480))
        self._identity = not any(d)
        self.Ks = torch.tensor(newK, dtype=torch.float32, device=self.device)
        self.gt_images = {}                                   # index -> undistorted uint8 image (the reference's tmp files)

    def __len__(self):
        return self.length

    def __getitem__(self, idx) -> Frame:
        if idx >= len(self):
            raise StopIteration
        im = np.array(self._Image.open(self.sequence_dir / self.rgb_frame_filenames[idx]).convert("RGB"))
        if not self._identity:
            im = remap_bilinear(im, self.undistort_map_x, self.undistort_map_y)
        x, y, w, h = self.roi
        im = im[y:y + h, x:x + w]
        self.gt_images[idx] = im
        image = torch.from_numpy(np.float32(im) / 255.0).to(self.device)
        dep = np.asarray(self._Image.open(self.sequence_dir / self.depth_frame_filenames[idx]))[y:y + h, x:x + w]
        depth = torch.from_numpy(dep.astype(np.float32)).to(self.device) / 5000.0
        height, width = image.shape[:2]
        cam = Camera(self.Ks.clone(), height, width)
        return Frame(image, float(self.rgb_frame_timestamps[idx]), cam, PoseZhou(torch.eye(4)).to(self.device),
                     torch.tensor(self.poses[idx], dtype=torch.float32), idx, gt_depth=depth,
                     img_file=str(self.sequence_dir / self.rgb_frame_filenames[idx]))
