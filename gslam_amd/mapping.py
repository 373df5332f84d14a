"""Keyframe bundle adjustment ("mapping") loop: the loss block, optimiser plumbing and iteration of
``Backend.optimize_map`` (gslam/backend.py:249-407, 554-602), without the process / viewer / rerun shell.

Multi-GPU: the window's keyframes are sharded over ranks (gslam_amd.dist.KeyframeShard); every rank holds a replica
of the map, renders its own cameras and everything that is summed over cameras travels in one RCCL all-reduce of a
single bucket (gslam_amd.dist.StepBucket) - on the launch-plan path (gslam_amd.plan.MappingStep).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import dist as gdist
from ._lib import check, lib, ptr, stream_ptr
from .losses import fused_mapping_loss, mapping_loss_and_grads
from .map import GaussianSplattingData
from .optim import FusedAdam, step_all
from .primitives import Frame
from .rasterization import RasterizationOutput
from .ssim import fused_ssim
from .utils import StopOnPlateau, create_batch, edge_aware_tv


@dataclass
class MapConfig:
    """Subset of gslam/backend.py:43-107 that the loss / optimiser path reads (same names, same defaults)."""
    isotropic_regularization_weight: float = 0.0005
    depth_regularization_weight: float = 0.000001
    pose_optim_lr: float = 0.003
    means_lr: float = 0.0016
    opacity_lr: float = 0.025
    scale_lr: float = 0.005
    color_lr: float = 0.01
    quat_lr: float = 0.005
    log_uncertainty_lr: float = 0.0025
    opacity_decay: float = 0.995
    optim_window_last_n_keyframes: int = 8
    num_iters_mapping: int = 15
    num_iters_initialization: int = 400
    ssim_weight: float = 0.2
    active_gs: bool = True
    enable_visibility_pruning: bool = False     # backend.py:94; the only reader of RasterizationOutput.n_touched
    device: str = 'cuda'


SPLAT_LRS = (('means', 'means_lr'), ('quats', 'quat_lr'), ('scales', 'scale_lr'), ('opacities', 'opacity_lr'),
             ('colors', 'color_lr'), ('log_uncertainties', 'log_uncertainty_lr'))


def mapping_loss(splats: GaussianSplattingData, outputs: RasterizationOutput, gt_imgs: torch.Tensor,
                 exposure_params: torch.Tensor, conf: MapConfig, regularize: bool = True, c_total: Optional[int] = None,
                 visible_gaussians: Optional[torch.Tensor] = None, iso_scale: float = 1.0):
    """gslam/backend.py:273-318.  ``c_total`` (cameras in the whole window) rescales the per-camera means when this
    rank holds only a shard: mean over C = sum over shards of (C_local / C) * local mean."""
    C_local = gt_imgs.shape[0]
    shard = 1.0 if c_total is None else C_local / float(c_total)
    rendered = outputs.rgbs * exposure_params[..., 0].view(-1, 1, 1, 1).exp() + exposure_params[..., 1].view(-1, 1, 1, 1)
    if conf.active_gs:
        photometric = (rendered - gt_imgs).square().sum(dim=-1)
        photometric = (photometric / (2 * outputs.betas.square())).mean()
        photometric = photometric + (outputs.betas.log().square() * 0.5).mean()
    else:
        photometric = (outputs.rgbs - gt_imgs).square().mean()
    if visible_gaussians is None:
        visible_gaussians = outputs.radii.sum(dim=0) > 0
    # same value as indexing with the boolean mask (backend.py:287-296) but without the nonzero() host sync
    mean_scales = splats.scales.mean(dim=1, keepdim=True).exp().detach()
    isotropic = ((splats.scales.exp() - mean_scales).abs() * visible_gaussians[:, None]).sum()
    depth_reg = edge_aware_tv(outputs.depthmaps, outputs.rgbs, outputs.alphas[..., 0] > 0.4)
    ssim_loss = 1.0 - fused_ssim(outputs.rgbs.permute(0, 3, 1, 2), gt_imgs.permute(0, 3, 1, 2), padding='valid')
    total = shard * ((1.0 - conf.ssim_weight) * photometric + conf.ssim_weight * ssim_loss) \
        + iso_scale * conf.isotropic_regularization_weight * isotropic
    if regularize:
        total = total + conf.depth_regularization_weight * depth_reg     # a SUM: no shard scaling
    return total, photometric


class MapOptimizers:
    """The six splat Adams + the pose Adam of backend.py:554-602,665-670 as two multi-tensor launches."""

    def __init__(self, splats: GaussianSplattingData, conf: MapConfig, capturable: bool = False):
        self.conf = conf
        self.capturable = capturable
        self.splat_opt = FusedAdam([{"params": [getattr(splats, name)], "lr": getattr(conf, lr)}
                                    for name, lr in SPLAT_LRS], capturable=capturable)
        self.pose_opt: Optional[FusedAdam] = None
        self._pose_ids = set()

    def add_pose(self, pose: torch.nn.Module, lr: Optional[float] = None):
        params = [p for p in pose.parameters() if p.requires_grad and id(p) not in self._pose_ids]
        if not params:
            return
        self._pose_ids.update(id(p) for p in params)
        group = {"params": params, "lr": self.conf.pose_optim_lr if lr is None else lr}
        if self.pose_opt is None:
            self.pose_opt = FusedAdam([group], capturable=self.capturable)
        else:
            self.pose_opt.add_param_group(group)

    def zero_grad(self):
        self.splat_opt.zero_grad(set_to_none=True)
        if self.pose_opt is not None:
            self.pose_opt.zero_grad(set_to_none=True)

    def step(self, decay=None) -> bool:
        """one launch for the step counters, one multi-tensor launch per (betas, eps) class for splats and poses
        together; returns whether ``decay`` (see optim.step_all) was applied by the update"""
        opts = [self.splat_opt] + ([self.pose_opt] if self.pose_opt is not None else [])
        return step_all(opts, decay)


class BundleAdjuster:
    """One object per process/GPU.  Two ways to run one iteration of the loop at backend.py:260-359 over a window:

    * ``step(window)`` - the reference-shaped path: ``splats(cameras, poses)`` through the autograd operators of this
      package, the fused loss, ``backward()``, fused Adam.  Single GPU; kept as the independent implementation the
      launch plan is tested against.
    * ``plan(window).step()`` - the production path: the same launches as a plan over persistent buffers replayed from a
      HIP graph (gslam_amd.plan.MappingStep), sharded over ranks when torch.distributed is initialised."""

    def __init__(self, splats: GaussianSplattingData, conf: Optional[MapConfig] = None, fused_loss: bool = True,
                 capturable: bool = False, need_n_touched: Optional[bool] = None, exchange_ranges: int = 0,
                 exchange_overlap: bool = True):
        """need_n_touched: keep the rasteriser's touched-pixel counts in ``last_outputs`` (read by visibility pruning
        only, backend.py:370-375); default = the configuration's ``enable_visibility_pruning`` (off, backend.py:94).
        capturable: device-side Adam step counters (needed by ``plan()``).
        exchange_ranges / exchange_overlap (more than one rank): the ranged, overlapped gradient / parameter exchange of
        gslam_amd.plan.MappingStep; 0 = the one-shot exchange (default until measured on a multi-GPU node)."""
        self.exchange_ranges, self.exchange_overlap = int(exchange_ranges), bool(exchange_overlap)
        self.splats = splats
        self.conf = conf or MapConfig()
        self.need_n_touched = bool(getattr(self.conf, "enable_visibility_pruning", False)) \
            if need_n_touched is None else bool(need_n_touched)
        self.fused_loss = fused_loss
        self.optimizers = MapOptimizers(splats, self.conf, capturable=capturable)
        self.shard = gdist.KeyframeShard()
        self.total_step = 0
        self.last_outputs: Optional[RasterizationOutput] = None
        self._plans: dict = {}

    def sync_moments(self):
        """Multi-GPU, before anything re-packs the map (pruning, insertion): the sharded update keeps each Adam moment valid
        on its owner only, and a re-pack moves rows between the owners' chunks - every rank gets both moments whole first (two
        all-gathers; a collective: all ranks call it at the same point of the loop).  No-op on one rank."""
        for p in self._plans.values():
            if getattr(p, "world", 1) > 1 and p.flat_state is not None and p.matches(self.splats, p.window):
                p.gather_moments()
                return

    def map_changed(self):
        """call after gslam_amd.pruning / gslam_amd.insertion re-packed the map (new parameter tensors, new N): launch
        plans and their captured graphs describe the old tensors and are dropped"""
        self._plans.clear()
        self.last_outputs = None

    def plan(self, window: List[Frame], regularize: bool = True, decay_opacity: bool = True):
        """the launch plan of one BA iteration over ``window`` (cached while the window's frames and the map's tensors stay
        the same objects); needs capturable=True"""
        from .plan import MappingStep
        key = (tuple(id(f) for f in window), bool(regularize), bool(decay_opacity))
        p = self._plans.get(key)
        if p is None or not p.matches(self.splats, window):
            if len(self._plans) >= 4:
                self._plans.clear()
            p = MappingStep(self.splats, self.optimizers, window, self.conf, regularize=regularize, shard=self.shard,
                            need_n_touched=self.need_n_touched, decay_opacity=decay_opacity,
                            exchange_ranges=self.exchange_ranges, exchange_overlap=self.exchange_overlap)
            self._plans[key] = p
        return p

    def step(self, window: List[Frame], regularize: bool = True, decay_opacity: bool = True):
        """reference-shaped iteration (autograd operators), single GPU"""
        total, photometric = self.render_backward(window, regularize)
        self.update(decay_opacity)
        return total, photometric

    def render_backward(self, window: List[Frame], regularize: bool = True):
        if self.shard.world_size > 1:
            raise RuntimeError("the autograd-shaped step is single-GPU; sharded BA runs on BundleAdjuster.plan()")
        conf = self.conf
        self.total_step += 1
        for f in window:
            self.optimizers.add_pose(f.pose)
        self.optimizers.zero_grad()
        cameras = [f.camera for f in window]
        poses = [f.pose for f in window]
        # the keyframes' images do not change between iterations: stack them once per window (the reference, and the
        # first version here, re-stacked 29 MB per iteration at 8 keyframes)
        key = tuple((f.img.data_ptr(), f.img._version) for f in window)
        cached = getattr(self, "_gt_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, create_batch(window, lambda f: f.img))
            self._gt_cache = cached
        gt_imgs = cached[1]
        exposure = create_batch(window, lambda f: f.exposure_params)
        outputs = self.splats(cameras, poses, render_depth=True, need_n_touched=self.need_n_touched)
        vis_count = outputs._vis_count                                  # = (radii > 0).sum(0), from K1
        # backend.py:326 means2d.retain_grad(): the rasteriser's backward hands the same values over as a view of its
        # gradient records (no copy kernel)
        outputs.means2d._gsx_share_grad = True
        if self.fused_loss:
            # value and analytic gradient in one pass (csrc/loss.hip); the backward is seeded at the render tensor and
            # runs as soon as that gradient exists, so that the isotropic term can be added into scales.grad in place.
            out2, v_render, v_exposure, v_scales = mapping_loss_and_grads(
                outputs, gt_imgs, exposure, self.splats.scales, ssim_weight=conf.ssim_weight,
                iso_weight=conf.isotropic_regularization_weight,
                tv_weight=conf.depth_regularization_weight if regularize else 0.0, active_gs=conf.active_gs,
                shard=1.0, vis_count=vis_count,
                backward_fn=lambda v: torch.autograd.backward([outputs._render], [v]),
                iso_grad_fn=lambda: self.splats.scales.grad)
            total, photometric = out2[0], out2[1]
            if v_scales is not None:
                self.splats.scales.grad.add_(v_scales)
            for i, f in enumerate(window):
                if f.exposure_params.requires_grad:
                    f.exposure_params.grad = v_exposure[i] if f.exposure_params.grad is None \
                        else f.exposure_params.grad + v_exposure[i]
        else:
            total, photometric = mapping_loss(self.splats, outputs, gt_imgs, exposure, conf, regularize,
                                              c_total=len(window), visible_gaussians=vis_count > 0)
            total.backward()
        self._vis_count = vis_count
        self.last_outputs = outputs
        return total.detach(), photometric.detach()

    def update(self, decay_opacity: bool = True):
        conf = self.conf
        vis_count = self._vis_count
        # backend.py:356-359: opacities of Gaussians seen by more than one camera decay after the update; the masked
        # multiply rides in the Adam launch when that launch is device-stepped
        op = self.splats.opacities
        decay = (op, vis_count.contiguous(), 1, conf.opacity_decay) if (decay_opacity and op.grad is not None) else None
        done = self.optimizers.step(decay) and decay is not None
        if decay_opacity and not done:
            with torch.no_grad():
                check(lib.gsx_opacity_decay(ptr(op.data), ptr(vis_count), op.shape[0], 1, conf.opacity_decay,
                                            stream_ptr(op.device)), "gsx_opacity_decay")

    def optimize_map(self, window: List[Frame], n_iters: Optional[int] = None, regularize: bool = True,
                     early_stop: bool = True):
        """backend.py:249-362 without pruning / insertion, on the autograd-shaped step."""
        n_iters = self.conf.num_iters_mapping if n_iters is None else n_iters
        stopper = StopOnPlateau(3, 0.012)
        last = None
        for _ in range(n_iters):
            total, photometric = self.step(window, regularize)
            last = (total, photometric)
            if early_stop and stopper.stop(photometric.item()):
                break
        outputs = self.last_outputs
        if outputs is not None:
            for f, d in zip(window, outputs.depthmaps):
                f.est_depths = d.detach().clone()
        return last


def optimize_poses_lbfgs(splats: GaussianSplattingData, window: List[Frame], conf: Optional[MapConfig] = None,
                         max_eval: Optional[int] = None):
    """``Backend.optimize_poses_lbfgs`` (gslam/backend.py:447-506): one strong-Wolfe L-BFGS step (history 10,
    tolerance_change 1e-7, torch defaults lr = 1, max_iter = 20) over the poses of the window with the photometric
    loss only; the pose of frame 0 stays fixed (:459-462), exposure parameters are not optimised (:463-464).  Every
    closure is one C-camera render forward + backward through the fused loss kernel.  Returns the last loss."""
    conf = conf or MapConfig()
    cameras = [x.camera for x in window]
    poses = [x.pose for x in window]
    gt_imgs = create_batch(window, lambda x: x.img)
    exposure = create_batch(window, lambda x: x.exposure_params).detach()
    params = []
    for x in window:
        if x.index == 0:
            continue
        params.extend(list(x.pose.parameters()))
    if not params:
        return None
    kw = {} if max_eval is None else {"max_eval": max_eval}
    optimizer = torch.optim.LBFGS(params, history_size=10, line_search_fn='strong_wolfe', tolerance_change=1e-7, **kw)
    last_loss = None
    frozen = [p.requires_grad for p in splats.parameters()]      # only the poses move here

    def closure():
        nonlocal last_loss
        if torch.is_grad_enabled():
            optimizer.zero_grad()
        outputs = splats(cameras, poses, render_depth=True, need_n_touched=False)
        # photometric term alone (backend.py:484-492): the fused block with the SSIM / isotropic / TV weights at zero
        _total, photometric = fused_mapping_loss(outputs, gt_imgs, exposure, None, ssim_weight=0.0, iso_weight=0.0,
                                                 tv_weight=0.0, active_gs=conf.active_gs)
        loss = _total
        if loss.requires_grad:
            loss.backward()
        last_loss = loss.item()                                  # backend.py:501
        return loss

    for p in splats.parameters():
        p.requires_grad_(False)
    try:
        optimizer.step(closure)
    finally:
        for p, r in zip(splats.parameters(), frozen):
            p.requires_grad_(r)
    return last_loss


class GraphedPoseRefiner:
    """`Backend.optimize_poses_lbfgs` (gslam/backend.py:447-506) with the optimiser on the device:
    torch.optim.LBFGS(history_size=10, strong_wolfe, tolerance_change=1e-7) over the poses of the window (the pose of
    frame 0 stays fixed, :459-462) restated as the state machine of csrc/track_opt.h (sized for 80 parameters in
    csrc/window_opt.hip) and advanced by one single-wave launch at the end of the closure.  The closure is a launch plan
    (gslam_amd.plan.WindowClosure: C <= 8 cameras against the frozen map, photometric term only, pose-only backward)
    over the plan's OWN pose / exposure / image slots, recorded into a HIP graph on the plan's own stream: one refinement
    is ``max_eval + 1`` replays and ONE read-back instead of a `.item()` per closure (backend.py:501).  The window's
    poses are copied into the slots before and out of them after a run, so the same refiner serves every window of the
    same shape (number of cameras, which of them are learnable, image size) over the same map tensors."""

    def __init__(self, splats: GaussianSplattingData, window: List[Frame], conf: Optional[MapConfig] = None,
                 max_eval: int = 25):
        from .plan import WindowClosure
        self.splats, self.conf, self.max_eval = splats, conf or MapConfig(), int(max_eval)
        self.window = list(window)
        learnable = self.learnable_flags(self.window)
        self.plan = WindowClosure(splats, [x.camera for x in self.window], learnable, active_gs=self.conf.active_gs)
        self.n = self.plan.n_params
        self.params = [p for x, l in zip(self.window, learnable) if l for p in x.pose.parameters()]
        self.loss = self.plan.out2[0:1]

    @staticmethod
    def learnable_flags(window: List[Frame]):
        return [bool(x.index != 0 and getattr(x.pose, "is_learnable", True)
                     and all(p.requires_grad for p in x.pose.parameters())) for x in window]

    def shape_key(self):
        c = self.plan
        return (c.C, tuple(c.slots.learnable), c.r.W, c.r.H, c.r.N)

    def matches(self, splats, window: List[Frame]) -> bool:
        """this refiner (and its captured graph) can serve ``window`` over ``splats`` as it is"""
        c = self.plan
        return (c.r.matches(splats) and len(window) == c.C and self.learnable_flags(window) == c.slots.learnable
                and all(int(x.camera.width) == c.r.W and int(x.camera.height) == c.r.H for x in window))

    @property
    def graph(self):
        return self.plan.graph if self.plan.graph.captured else None

    def capture(self):
        from ._sync import capture_lock
        with capture_lock:
            self.plan.prepare()

    def _issue(self):
        self.plan.init_optimizer(self.max_eval)
        self.plan.launch(self.max_eval + 1)

    def run_async(self, window: Optional[List[Frame]] = None):
        """one refinement without any read-back (the caller polls ``capacity_ok()`` later): the graph must have been
        captured by an earlier run()"""
        window = self.window if window is None else list(window)
        assert self.plan.graph.captured
        self.plan.load(window)
        self._issue()
        self.plan.store(window)

    def capacity_ok(self) -> bool:
        return self.plan.r.check_capacity()

    def run(self, window: Optional[List[Frame]] = None):
        """-> (last closure loss, number of closure evaluations); the poses of ``window`` (default: the window given at
        construction) are updated in place"""
        window = self.window if window is None else list(window)
        for _attempt in range(3):
            self.plan.load(window)
            if not self.plan.graph.captured or self.plan.r.stale:
                self.capture()
            self._issue()
            rep = self.plan.read_report().cpu()
            if self.plan.r.check_capacity():
                self.plan.store(window)
                return float(rep[4]), int(rep[1])
            # the tile lists outgrew the captured capacity: buffers were grown, re-capture and redo from the saved poses
        raise RuntimeError("intersection buffers kept overflowing")


class GraphedBundleAdjuster:
    """A BA iteration over a FIXED window replayed from HIP graphs: thin facade over ``BundleAdjuster.plan(window)``
    (gslam_amd.plan.MappingStep) - pose chain, K1, binning, sort, K8, SSIM, loss, K9, K2, Adam, decay are one graph launch
    on the host.  Multi-GPU: two graphs (render + loss + backward | isotropic + Adam + decay) around the one eager
    all-reduce.  Needs a BundleAdjuster built with capturable=True; the window's poses are updated in place, its images
    and exposure parameters are constants of the plan (``plan.refresh_inputs()`` after changing them).
    ``warmup`` plan iterations run eagerly (real updates) before the capture.  The tile-list capacity is baked into the graph
    and the update is gated on the device by the render's overflow status: after an overflow every further ``step()``
    applies NOTHING until ``capacity_ok()`` has been polled (it grows the lists; the next step re-captures) - poll it after
    every step, or after every short run of steps and redo that run when it returns False (``step_checked()`` does both)."""

    def __init__(self, ba: BundleAdjuster, window: List[Frame], warmup: int = 0, regularize: bool = True,
                 decay_opacity: bool = True):
        assert ba.optimizers.capturable, "build the BundleAdjuster with capturable=True"
        self.ba, self.window = ba, window
        self.plan = ba.plan(window, regularize, decay_opacity)
        self.multi = self.plan.world > 1
        for _ in range(int(warmup)):
            self.plan.step(graphed=False)
            ba.total_step += 1
        self.plan.prepare()
        self.total, self.photometric = self.plan.out2[0], self.plan.out2[1]

    def step(self):
        self.plan.step()
        self.ba.total_step += 1
        return self.total, self.photometric

    def step_checked(self, max_attempts: int = 4):
        """step() + the read-back of loss and overflow flag (the reference's loss.item(), backend.py:351), redone while the
        tile lists overflow -> (total, photometric) as floats"""
        for _ in range(max_attempts):
            self.plan.step()
            total, pm, ok = self.plan.finish_step()
            if ok:
                self.ba.total_step += 1
                return total, pm
        raise RuntimeError(f"BA iteration still overflows its tile lists after {max_attempts} attempts")

    def capacity_ok(self) -> bool:
        return self.plan.capacity_ok()
