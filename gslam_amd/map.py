"""GaussianSplattingData: the plugin boundary object of gslam/map.py:13-164.  ``__call__(cameras, poses,
render_depth)`` renders through gslam_amd.rasterization with exactly the live argument set of map.py:88-103."""
from __future__ import annotations

from typing import List

import torch

from .primitives import Camera, Pose, pose_batch
from .rasterization import RasterizationOutput, rasterization
from .utils import create_batch


class GaussianSplattingData(torch.nn.Module):
    _per_splat_params = ['means', 'quats', 'scales', 'opacities', 'colors', 'log_uncertainties', 'ages']

    def __init__(self, means, quats, scales, opacities, colors, log_uncertainties, ages):
        super().__init__()
        self.means = torch.nn.Parameter(means)
        self.quats = torch.nn.Parameter(quats)
        self.scales = torch.nn.Parameter(scales)
        self.opacities = torch.nn.Parameter(opacities)
        self.colors = torch.nn.Parameter(colors)
        self.log_uncertainties = torch.nn.Parameter(log_uncertainties)
        self.ages = torch.nn.Parameter(ages, requires_grad=False)
        self.register_buffer('background', torch.tensor([0.0, 0.0, 0.0], device=self.means.device).float())

    def _render(self, cameras: List[Camera], viewmats: torch.Tensor, render_mode: str, visibility_min_T: float,
                need_n_touched: bool = True, capacity=None):
        Ks = create_batch(cameras, lambda x: x.intrinsics)
        return rasterization(
            means=self.means, quats=self.quats, log_scales=self.scales, logit_opacities=self.opacities,
            logit_colors=self.colors, viewmats=viewmats, Ks=Ks, width=cameras[0].width, height=cameras[0].height,
            render_mode=render_mode, packed=False, log_uncertainties=self.log_uncertainties,
            visibility_min_T=visibility_min_T, backgrounds=self._backgrounds(len(cameras)),
            need_n_touched=need_n_touched, capacity=capacity)

    def _backgrounds(self, n_cams: int) -> torch.Tensor:
        """self.background.tile([C, 1]) (map.py:73,102), cached per (C, buffer version)"""
        key = (n_cams, self.background._version, self.background.data_ptr())
        cache = self.__dict__.setdefault("_bg_cache", {})
        if key not in cache:
            while len(cache) >= 8:
                cache.pop(next(iter(cache)))
            cache[key] = self.background.tile([n_cams, 1])
        return cache[key]

    def render(self, cameras: List[Camera], viewmats: List[torch.Tensor], visibility_min_T: float = 0.5):
        return self._render(cameras, create_batch(viewmats), 'RGB+D', visibility_min_T)

    def forward(self, cameras: List[Camera], poses: List[Pose], render_depth: bool = False,
                visibility_min_T: float = 0.5, need_n_touched: bool = True, capacity=None) -> RasterizationOutput:
        """``need_n_touched=False`` (extension): skip the touched-pixel counts nobody reads in the optimisation loops
        (only visibility pruning does, backend.py:370-375); ``outputs.n_touched`` is then None.  ``capacity``
        (extension): a gslam_amd.rasterization.IsectCapacity for sync-free renders of a repeated shape."""
        viewmats = pose_batch(poses)                                     # = create_batch(poses, lambda x: x())
        return self._render(cameras, viewmats, 'RGB+D' if render_depth else 'RGB', visibility_min_T, need_n_touched,
                            capacity)

    @staticmethod
    def empty(device: str = 'cuda') -> "GaussianSplattingData":
        e = lambda: torch.tensor([], device=device)
        return GaussianSplattingData(e(), e(), e(), e(), e(), e(), e().long())

    @staticmethod
    def from_dict(d: dict, device=None) -> "GaussianSplattingData":
        g = lambda k: d[k] if device is None else d[k].to(device)
        return GaussianSplattingData(g('means'), g('quats'), g('scales'), g('opacities'), g('colors'),
                                     g('log_uncertainties'), g('ages'))

    def clone(self) -> "GaussianSplattingData":
        return GaussianSplattingData(*[getattr(self, p).clone().detach() for p in self._per_splat_params])

    def mask(self, m) -> "GaussianSplattingData":
        return GaussianSplattingData(*[getattr(self, p)[m] for p in self._per_splat_params])

    def no_grad_clone(self) -> "GaussianSplattingData":
        ret = self.clone()
        for p in self._per_splat_params:
            getattr(ret, p).requires_grad_(False)
        return ret

    def as_dict(self):
        return torch.nn.ParameterDict({p: getattr(self, p) for p in self._per_splat_params})
