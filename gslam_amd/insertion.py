"""Map growth between bundle-adjustment iterations: gslam/insertion.py (same class names, arguments, return values).

The per-array ``torch.cat`` / boolean-indexing kernels of ``_add_new_splats`` / ``_duplicate`` / ``_split`` are one
launch each over all per-Gaussian arrays (csrc/maintain.hip through gslam_amd.pruning.gather_rows / concat_rows); the
covariance of the split sampling is the ``quat_scale_to_covar_preci`` kernel (insertion.py:88-91); the occlusion test
of new splats against the keyframes' depth maps uses the packed projection (``get_new_splat_depth``,
rasterization.py:363-448).  Sampling decisions (random pixels, noise) use torch's device RNG like the reference."""
from __future__ import annotations

import math
from abc import ABC
from typing import Dict, List, Optional

import torch

from . import ops
from .map import GaussianSplattingData
from .primitives import Frame
from .pruning import _optimizer_of, _rebuild, _state_tensors, concat_rows, gather_rows
from .rasterization import RasterizationOutput, get_new_splat_depth


def knn(x: torch.Tensor, K: int = 4) -> torch.Tensor:
    """distances to the K nearest neighbours (self included), gslam/utils.py:26-30, on the device"""
    d = torch.cdist(x, x)
    return d.topk(min(K, x.shape[0]), dim=-1, largest=False).values


class InsertionStrategy(ABC):
    def step(self, splats: GaussianSplattingData, optimizers, rasterization_output: RasterizationOutput, frame: Frame,
             N: int):
        return

    @torch.no_grad()
    def _add_new_splats(self, splats: GaussianSplattingData, optimizers, new_params: Dict[str, torch.Tensor]) -> int:
        """gslam/insertion.py:27-63: append the new rows to every parameter, zero rows to its Adam moments"""
        N = int(new_params['means'].shape[0])
        if N == 0:
            return 0
        if splats.means.shape[0] == 0:
            # first insertion into GaussianSplattingData.empty() (backend.py:604-637): the new rows ARE the map
            new_values = {name: new_params[name].detach().to(p.dtype).clone() for name, p in splats.named_parameters()}
            _rebuild(splats, optimizers, new_values, {})
            return N
        a, b, slots = [], [], []
        for name, p in splats.named_parameters():
            a.append(p.data)
            b.append(new_params[name].reshape((N,) + tuple(p.shape[1:])))
            slots.append(("param", name, None))
            opt, _gi = _optimizer_of(optimizers, name)
            if opt is not None:
                for key, val in _state_tensors(opt, p).items():
                    a.append(val)
                    b.append(None)                                  # append zero here (insertion.py:52-56)
                    slots.append(("state", name, key))
        outs = concat_rows(a, b, N)
        new_values, new_states = {}, {}
        for (kind, name, key), t in zip(slots, outs):
            if kind == "param":
                new_values[name] = t
            else:
                new_states.setdefault(name, {})[key] = t
        _rebuild(splats, optimizers, new_values, new_states)
        return N

    @torch.no_grad()
    def _duplicate(self, splats: GaussianSplattingData, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
        """gslam/insertion.py:65-74"""
        index = torch.nonzero(mask, as_tuple=False).reshape(-1)
        names = [n for n, _ in splats.named_parameters()]
        outs = gather_rows([p.data for _, p in splats.named_parameters()], index)
        return dict(zip(names, outs))

    @torch.no_grad()
    def _split(self, splats: GaussianSplattingData, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
        """gslam/insertion.py:76-97: sample the new mean from the Gaussian itself, shrink the scales by 1.6"""
        ret = self._duplicate(splats, mask)
        covars, _precis = ops.quat_scale_to_covar_preci(ret['quats'], torch.exp(ret['scales']))
        noise = torch.randn_like(ret['means'])
        noise = torch.einsum("bij,bj->bi", covars, noise)
        ret['means'].add_(noise)
        ret['scales'].add_(-math.log(1.6))
        return ret


class InsertFromDepthMap(InsertionStrategy):
    """gslam/insertion.py:100-284"""

    def __init__(self, depth_variance: float, no_depth_variance: float, min_alpha_for_depth: float,
                 initial_opacity: float, insert_in_regions_with_depth: bool = True, global_pause_event=None):
        self.depth_variance = depth_variance
        self.no_depth_variance = no_depth_variance
        self.min_alpha_for_depth = min_alpha_for_depth
        self.initial_opacity = initial_opacity
        self.insert_in_regions_with_depth = insert_in_regions_with_depth
        self.global_pause_event = global_pause_event

    @torch.no_grad()
    def step(self, splats: GaussianSplattingData, optimizers, rasterization_output: RasterizationOutput, frame: Frame,
             N: int, keyframes: List[Frame], gt_depthmap: Optional[torch.Tensor] = None):
        depths = (rasterization_output.depthmaps[0, ...] if gt_depthmap is None else gt_depthmap).clone()
        alphas = rasterization_output.alphas[0, ..., 0]
        device = depths.device
        valid = torch.logical_and(alphas > self.min_alpha_for_depth, depths > 0)
        n_valid = int(valid.sum().item())
        n_invalid = depths.numel() - n_valid
        # prefer to add N splats in the region where we don't have geometry already (insertion.py:143-147)
        n_invalid_splats = min(N, n_invalid)
        n_valid_splats = max(0, min(int(N / 2) - n_invalid_splats, n_valid))
        if n_invalid_splats <= 0 and (not self.insert_in_regions_with_depth and n_valid_splats <= 0):
            return 0
        median_depth = depths[valid].median() if n_valid > 0 else depths.median()
        noise = torch.randn_like(depths)
        depths = torch.where(valid, depths + noise * self.depth_variance,
                             median_depth + noise * self.no_depth_variance)
        depths.clamp_min_(0.1)
        idx_valid = torch.nonzero(valid.reshape(-1)).reshape(-1)
        idx_invalid = torch.nonzero(~valid.reshape(-1)).reshape(-1)
        picks = []
        if n_invalid_splats > 0:
            picks.append(idx_invalid[torch.randint(idx_invalid.shape[0], [n_invalid_splats], device=device)])
        if self.insert_in_regions_with_depth and n_valid_splats > 0:
            picks.append(idx_valid[torch.randint(idx_valid.shape[0], [n_valid_splats], device=device)])
        if len(picks) == 0:
            return 0
        picks = torch.cat(picks)
        N = int(picks.shape[0])
        means = frame.camera.backproject(depths)[picks]
        colors = frame.img.reshape([-1, 3])[picks]
        c2w = torch.linalg.inv(frame.pose().detach())
        means = means @ c2w[:3, :3].t() + c2w[:3, 3]
        if splats.scales.numel() > 0:
            scales = torch.exp(splats.scales.detach()).median(dim=0)[0].tile([N, 1])
        else:
            nn3 = torch.sqrt(knn(means, 4)[:, 1:] ** 2).mean(dim=-1)
            scales = nn3.unsqueeze(-1).repeat(1, 3)
        new_params = {
            'means': means.float(),
            'scales': torch.log(scales),
            'colors': torch.logit(colors),
            'opacities': torch.logit(torch.full((N,), self.initial_opacity, device=device)),
            'quats': torch.rand((N, 4), device=device),
            'log_uncertainties': torch.ones((N,), device=device),
            'ages': torch.full((N,), frame.index, device=device).long(),
        }
        if len(keyframes) > 1:
            # drop new splats that would sit in front of a keyframe's estimated surface (insertion.py:245-277)
            Ks = torch.stack([x.camera.intrinsics for x in keyframes])
            viewmats = torch.stack([x.pose().detach() for x in keyframes])
            est_depths = torch.stack([x.est_depths for x in keyframes])
            height, width = frame.camera.height, frame.camera.width
            camera_ids, gaussian_ids, _, means2d, d_new = get_new_splat_depth(new_params, viewmats, Ks, width, height)
            m2 = means2d.to(torch.int64)
            mw = torch.clamp(m2[:, 0], max=width - 1, min=0)
            mh = torch.clamp(m2[:, 1], max=height - 1, min=0)
            in_front = d_new < est_depths[camera_ids, mh, mw]
            keep = torch.ones(N, dtype=torch.bool, device=device)
            keep[gaussian_ids[in_front]] = False
            names = list(new_params.keys())
            kept = gather_rows([new_params[k] for k in names], torch.nonzero(keep).reshape(-1))
            new_params = dict(zip(names, kept))
        return self._add_new_splats(splats, optimizers, new_params)


class InsertUsingImagePlaneGradients(InsertionStrategy):
    """The densification of the original 3DGS paper, gslam/insertion.py:287-347"""

    def __init__(self, grow_grad2d: float, grow_scale3d: float):
        self.grow_grad2d = grow_grad2d
        self.grow_scale3d = grow_scale3d

    @torch.no_grad()
    def step(self, splats: GaussianSplattingData, optimizers, rasterization_output: RasterizationOutput, frame: Frame,
             N: int, window_cameras: Optional[int] = None, reduce_sum=None):
        """window_cameras / reduce_sum (keyframe-sharded BA): ``rasterization_output`` then holds only this rank's cameras of
        a window of ``window_cameras``; the per-Gaussian statistic - a mean over the window's cameras - is summed over
        ranks with ``reduce_sum`` (an all-reduce) so that every replica densifies the same Gaussians"""
        n_cams = rasterization_output.n_cameras if window_cameras is None else int(window_cameras)
        grads = rasterization_output.means2d.grad.clone()
        # normalize grads by image size (insertion.py:300-306)
        grads[..., 0] *= rasterization_output.width / 2.0 * n_cams
        grads[..., 1] *= rasterization_output.height / 2.0 * n_cams
        if reduce_sum is None and window_cameras is None:
            grads = grads.norm(dim=-1).mean(dim=0)
        else:
            grads = grads.norm(dim=-1).sum(dim=0)
            if reduce_sum is not None:
                reduce_sum(grads)
            grads = grads / n_cams
        high = grads > self.grow_grad2d
        is_small = torch.exp(splats.scales.detach()).max(dim=-1).values <= self.grow_scale3d
        to_duplicate = high & is_small
        to_split = high & ~is_small
        num_split = int(to_split.sum().item())
        num_duplicate = int(to_duplicate.sum().item())
        duplicated = split = None
        if num_duplicate > 0:
            duplicated = self._duplicate(splats, mask=to_duplicate)
            duplicated['log_uncertainties'].fill_(1.0)
        if num_split > 0:
            split = self._split(splats, mask=to_split)
            split['log_uncertainties'].fill_(1.0)
        if duplicated is not None:
            self._add_new_splats(splats, optimizers, duplicated)
        if split is not None:
            self._add_new_splats(splats, optimizers, split)
        r = rasterization_output.radii
        rasterization_output.radii = torch.cat(
            [r, torch.zeros([r.shape[0], num_duplicate + num_split], device=r.device, dtype=r.dtype)], dim=1)
        return num_duplicate, num_split


class SequentialInsertion(InsertionStrategy):
    """gslam/insertion.py:350-369"""

    def __init__(self, strategies: List[InsertionStrategy]):
        self.strategies = strategies

    def step(self, splats: GaussianSplattingData, optimizers, rasterization_output: RasterizationOutput, frame: Frame,
             N: int):
        for strategy in self.strategies:
            strategy.step(splats, optimizers, rasterization_output, frame, N)
