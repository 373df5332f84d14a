"""Trajectory evaluation of gslam/trajectory.py:14-97: similarity alignment (Kabsch-Umeyama) and
the absolute trajectory error the reference reports (`ate_*`).  Host-side numpy, as in the reference; plotting is left
out (viewer shell)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np


def kabsch_umeyama(A: np.ndarray, B: np.ndarray) -> Tuple[np.ndarray, float, np.ndarray]:
    """(R, c, t) with t + c * R @ b ~ a in the least-squares sense, A, B: [n, m] (trajectory.py:14-45; falls back to
    the identity when the SVD does not converge, as the reference does)."""
    assert A.shape == B.shape
    n, m = A.shape
    EA, EB = A.mean(axis=0), B.mean(axis=0)
    var_a = np.mean(np.linalg.norm(A - EA, axis=1) ** 2)
    try:
        H = ((A - EA).T @ (B - EB)) / n
        U, D, VT = np.linalg.svd(H)
        d = np.sign(np.linalg.det(U) * np.linalg.det(VT))
        S = np.diag([1.0] * (m - 1) + [d])
        R = U @ S @ VT
        c = var_a / np.trace(np.diag(D) @ S)
        t = EA - c * R @ EB
    except np.linalg.LinAlgError:
        R, c, t = np.eye(m), 1.0, np.zeros(m, dtype=np.float32)
    return R, float(c), t


def align(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """B mapped into A's frame"""
    R, c, t = kabsch_umeyama(A, B)
    return t + c * (B @ R.T)


def average_translation_error(A: np.ndarray, B: np.ndarray) -> float:
    """mean Euclidean distance after similarity alignment (trajectory.py:48-53)"""
    err = align(A, B) - A
    return float(np.mean(np.sqrt(np.sum(err * err, axis=-1))))


def evaluate_trajectories(trajectories: Dict[str, List], keyframe_indices: Optional[List[int]] = None) -> Dict[str, float]:
    """{'ate_<name>': ...} for every trajectory with at least two frames (trajectory.py:74-107 without the figure).
    A frame needs ``gt_pose`` ([4,4] tensor) and a callable ``pose`` returning the estimated [4,4] view matrix."""
    ates = {}
    for name, frames in trajectories.items():
        if len(frames) < 2:
            continue
        gt = np.array([f.gt_pose.detach().cpu().numpy() for f in frames])[:, :3, 3]
        est = np.array([f.pose().detach().cpu().numpy() for f in frames])[:, :3, 3]
        ates['ate_' + name] = average_translation_error(gt, align(gt, est))
    return ates
