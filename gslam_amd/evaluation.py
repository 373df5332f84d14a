"""Reconstruction metrics of gslam/frontend.py:374-409: the reference scores every rendered frame
against the (undistorted) input image with scikit-image's `peak_signal_noise_ratio` and `structural_similarity` on uint8
RGB and reports the means.  scikit-image is not a dependency here; both metrics are restated from their published
definitions with scikit-image's defaults (7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, border of
(win - 1) / 2 pixels cropped, data range 255 for uint8)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
from scipy.ndimage import uniform_filter


def psnr_uint8(img: np.ndarray, ref: np.ndarray) -> float:
    """skimage.metrics.peak_signal_noise_ratio(ref, img) for uint8 inputs (data_range 255)"""
    err = np.mean((img.astype(np.float64) - ref.astype(np.float64)) ** 2)
    return float('inf') if err == 0 else float(10.0 * np.log10(255.0 ** 2 / err))


def ssim_uint8(img: np.ndarray, ref: np.ndarray, win: int = 7) -> float:
    """skimage.metrics.structural_similarity(img, ref, channel_axis=2) for uint8 [H,W,C] inputs"""
    assert img.shape == ref.shape and img.ndim == 3
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    pad = (win - 1) // 2
    vals = []
    for ch in range(img.shape[2]):
        x, y = img[..., ch].astype(np.float64), ref[..., ch].astype(np.float64)
        ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
        uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def to_uint8(rgb) -> np.ndarray:
    """torch [H,W,3] in [0,1] -> uint8 as gslam/utils.py torch_image_to_np does (clamp, x255, truncate)"""
    a = rgb.detach().clamp(0.0, 1.0).mul(255.0).cpu().numpy()
    return a.astype(np.uint8)


def evaluate_reconstruction(splats, frames: List, gt_images: List[np.ndarray]) -> Dict[str, float]:
    """{'ssim': mean, 'psnr': mean} of the renders at the frames' final poses against their uint8 ground-truth images"""
    psnrs, ssims = [], []
    for f, gt in zip(frames, gt_images):
        out = splats([f.camera], [f.pose], render_depth=True)
        rgb = to_uint8(out.rgbs[0])
        psnrs.append(psnr_uint8(rgb, gt))
        ssims.append(ssim_uint8(rgb, gt))
    return {'ssim': float(np.mean(ssims)), 'psnr': float(np.mean(psnrs))}
