"""Deterministic synthetic scenes / cameras of TUM shape.

No dataset ships with the build: configs ①-⑤ of BASELINE.json are driven by these generators.  Everything is
generated on the CPU with a seeded ``torch.Generator`` and copied to the device by the caller, so that the CPU
oracle and the HIP path see bit-identical inputs.
"""
from __future__ import annotations

import math

import torch

TUM_K_640 = ((525.0, 0.0, 319.5), (0.0, 525.0, 239.5), (0.0, 0.0, 1.0))  # cf. gslam/utils.py:46-58
HD_K_1080 = ((1575.0, 0.0, 959.5), (0.0, 1575.0, 539.5), (0.0, 0.0, 1.0))


def _logit(p: torch.Tensor) -> torch.Tensor:
    return torch.log(p) - torch.log1p(-p)


def make_scene(n: int, seed: int = 0, sh_degree: int | None = None, dtype=torch.float32) -> dict:
    """Random Gaussian map in the parameterisation of gslam/map.py:14-43 (pre-activation values)."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    means = torch.stack([u(n) * 6.0 - 3.0, u(n) * 6.0 - 3.0, u(n) * 5.5 + 0.5], -1)
    quats = torch.randn(n, 4, generator=g, dtype=torch.float64)
    log_scales = torch.log(u(n, 3) * 0.045 + 0.005)
    logit_opacities = _logit(u(n) * 0.8 + 0.1)
    logit_colors = _logit(u(n, 3) * 0.9 + 0.05)
    log_uncertainties = torch.ones(n, dtype=torch.float64)  # gslam/insertion.py:242
    out = dict(means=means, quats=quats, scales=log_scales, opacities=logit_opacities, colors=logit_colors,
               log_uncertainties=log_uncertainties)
    if sh_degree is not None:
        k = (sh_degree + 1) ** 2
        out["sh_coeffs"] = torch.randn(n, k, 3, generator=g, dtype=torch.float64) * 0.2
    out = {k_: v.to(dtype).contiguous() for k_, v in out.items()}
    out["ages"] = torch.zeros(n, dtype=torch.int64)
    return out


def make_intrinsics(width: int = 640, height: int = 480, dtype=torch.float32) -> torch.Tensor:
    if (width, height) == (640, 480):
        return torch.tensor(TUM_K_640, dtype=dtype)
    if (width, height) == (1920, 1080):
        return torch.tensor(HD_K_1080, dtype=dtype)
    f = 525.0 * width / 640.0
    return torch.tensor(((f, 0.0, (width - 1) / 2.0), (0.0, f, (height - 1) / 2.0), (0.0, 0.0, 1.0)), dtype=dtype)


def sequence_param(i: int, period: int = 120) -> float:
    """pose parameter of frame ``i`` of the synthetic TUM-shape SEQUENCE: a sweep back and forth over the pose range of an
    8-keyframe window, c(i) = 4 + 4 sin(2 pi i / period) - at most ~1 cm and ~0.2 degrees between consecutive frames (hand-held
    motion at 30 Hz) and, unlike c = i, the camera keeps looking at the scene however long the sequence is"""
    return 4.0 + 4.0 * math.sin(2.0 * math.pi * i / period)


def make_viewmat(c: float, dtype=torch.float32, noise: torch.Tensor | None = None) -> torch.Tensor:
    """world->camera pose with parameter ``c``: 0.05*c m along x and 1 deg * c of yaw (c = 0..7 are the
    keyframes of a BA window)."""
    yaw = math.radians(1.0) * c
    cy, sy = math.cos(yaw), math.sin(yaw)
    R = torch.tensor(((cy, 0.0, -sy), (0.0, 1.0, 0.0), (sy, 0.0, cy)), dtype=torch.float64)
    pos = torch.tensor((0.05 * c, 0.0, 0.0), dtype=torch.float64)
    V = torch.eye(4, dtype=torch.float64)
    V[:3, :3] = R
    V[:3, 3] = -R @ pos
    if noise is not None:
        V[:3, 3] += noise.to(torch.float64)
    return V.to(dtype)


def make_cameras(n_cams: int, width: int = 640, height: int = 480, dtype=torch.float32, start: int = 0):
    Ks = make_intrinsics(width, height, dtype)[None].repeat(n_cams, 1, 1).contiguous()
    viewmats = torch.stack([make_viewmat(start + c, dtype) for c in range(n_cams)], 0).contiguous()
    return viewmats, Ks
