"""Keyframe bundle adjustment ("mapping") loop: the loss block, optimiser plumbing and iteration of
``Backend.optimize_map`` (gslam/backend.py:249-407, 554-602), without the process / viewer / rerun shell.

Multi-GPU: the window's keyframes are sharded over ranks (gslam_amd.dist.KeyframeShard); every rank holds a replica
of the map, renders its own cameras and the per-Gaussian gradients are summed with one RCCL all-reduce of a single
[N,15] bucket.  With world_size == 1 the code path is identical minus the collective.
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
    """One object per process/GPU.  ``step(window)`` is one iteration of the loop at backend.py:260-359."""

    def __init__(self, splats: GaussianSplattingData, conf: Optional[MapConfig] = None, fused_loss: bool = True,
                 capturable: bool = False, need_n_touched: Optional[bool] = None):
        """need_n_touched: keep the rasteriser's touched-pixel counts in ``last_outputs`` (read by visibility pruning
        only, backend.py:370-375); default = the configuration's ``enable_visibility_pruning`` (off, backend.py:94)."""
        self.splats = splats
        self.conf = conf or MapConfig()
        self.need_n_touched = bool(getattr(self.conf, "enable_visibility_pruning", False)) \
            if need_n_touched is None else bool(need_n_touched)
        self.fused_loss = fused_loss
        self.optimizers = MapOptimizers(splats, self.conf, capturable=capturable)
        self.shard = gdist.KeyframeShard()
        self.bucket = gdist.GradBucket(splats) if self.shard.world_size > 1 else None
        self.total_step = 0
        self.last_outputs: Optional[RasterizationOutput] = None

    def map_changed(self):
        """call after gslam_amd.pruning / gslam_amd.insertion re-packed the map (new parameter tensors, new N): the
        multi-GPU gradient bucket is rebuilt for the new size; captured graphs of the old map must be discarded."""
        if self.bucket is not None:
            self.bucket = gdist.GradBucket(self.splats)
        self.last_outputs = None

    def step(self, window: List[Frame], regularize: bool = True, decay_opacity: bool = True):
        """window = ALL keyframes of the BA window (every rank passes the same list); this rank renders its shard.
        = render_backward() -> reduce() -> update(); the three phases are separately callable so that the two
        compute phases can be replayed from HIP graphs around the one eager collective."""
        total, photometric = self.render_backward(window, regularize)
        self.reduce()
        self.update(decay_opacity)
        return total, photometric

    def render_backward(self, window: List[Frame], regularize: bool = True):
        conf = self.conf
        self.total_step += 1
        mine = self.shard.select(window)
        multi = self.shard.world_size > 1
        for f in mine:
            self.optimizers.add_pose(f.pose)
        self.optimizers.zero_grad()
        if self.bucket is not None:
            self.bucket.attach_zeroed()
        cameras = [f.camera for f in mine]
        poses = [f.pose for f in mine]
        # the keyframes' images do not change between iterations: stack them once per window (the reference, and the
        # first version here, re-stacked 29 MB per iteration at 8 keyframes)
        key = tuple((f.img.data_ptr(), f.img._version) for f in mine)
        cached = getattr(self, "_gt_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, create_batch(mine, lambda f: f.img))
            self._gt_cache = cached
        gt_imgs = cached[1]
        exposure = create_batch(mine, lambda f: f.exposure_params)
        outputs = self.splats(cameras, poses, render_depth=True, need_n_touched=self.need_n_touched)
        vis_count = outputs._vis_count                                  # = (radii > 0).sum(0), from K1
        # backend.py:326 means2d.retain_grad(): the rasteriser's backward hands the same values over as a view of its
        # gradient records (no copy kernel)
        outputs.means2d._gsx_share_grad = True
        if self.fused_loss:
            # value and analytic gradient in one pass (csrc/loss.hip); the backward is seeded at the render tensor and
            # runs as soon as that gradient exists, so that the isotropic term can be added into scales.grad in place.
            # Multi-GPU: the isotropic term needs the window-wide visibility, so it moves behind the all-reduce.
            out2, v_render, v_exposure, v_scales = mapping_loss_and_grads(
                outputs, gt_imgs, exposure, self.splats.scales, ssim_weight=conf.ssim_weight,
                iso_weight=0.0 if multi else conf.isotropic_regularization_weight,
                tv_weight=conf.depth_regularization_weight if regularize else 0.0, active_gs=conf.active_gs,
                shard=len(mine) / float(len(window)), vis_count=vis_count,
                backward_fn=lambda v: torch.autograd.backward([outputs._render], [v]),
                iso_grad_fn=lambda: self.splats.scales.grad)
            total, photometric = out2[0], out2[1]
            if v_scales is not None:
                self.splats.scales.grad.add_(v_scales)
            for i, f in enumerate(mine):
                if f.exposure_params.requires_grad:
                    f.exposure_params.grad = v_exposure[i] if f.exposure_params.grad is None \
                        else f.exposure_params.grad + v_exposure[i]
        else:
            assert not multi, "the torch-formulated loss is the single-GPU reference path"
            total, photometric = mapping_loss(self.splats, outputs, gt_imgs, exposure, conf, regularize,
                                              c_total=len(window), visible_gaussians=vis_count > 0)
            total.backward()
        if self.bucket is not None:
            self.bucket.counts.copy_(vis_count)                         # int32 -> fp32 column of the bucket
        self._vis_count = vis_count
        self.last_outputs = outputs
        return total.detach(), photometric.detach()

    def reduce(self):
        """the ONE data-path collective of an iteration: sum of the [N*15 + N] bucket over ranks (RCCL / xGMI)"""
        if self.bucket is not None:
            self.bucket.all_reduce()

    def update(self, decay_opacity: bool = True):
        conf = self.conf
        vis_count = self._vis_count
        if self.bucket is not None:
            vis_count = self.bucket.counts.to(torch.int32)              # window-wide visible-camera counts
            w = conf.isotropic_regularization_weight
            if w != 0.0:                                                # identical on every rank (replicated map)
                sc = self.splats.scales
                N = sc.shape[0]
                from .ops import workspace
                ws = workspace(lib.gsx_isotropic_workspace_bytes(N), sc.device, "iso")
                check(lib.gsx_isotropic_loss_acc(ptr(sc.data), ptr(vis_count), N, w, None, ptr(sc.grad), ptr(ws),
                                                 ws.numel(), stream_ptr(sc.device)), "gsx_isotropic_loss_acc")
        # backend.py:356-359: opacities of Gaussians seen by more than one camera decay after the update; the masked
        # multiply rides in the Adam launch when that launch is device-stepped (graph-capturable)
        op = self.splats.opacities
        decay = (op, vis_count.contiguous(), 1, conf.opacity_decay) if (decay_opacity and op.grad is not None) else None
        done = self.optimizers.step(decay) and decay is not None
        if decay_opacity and not done:
            with torch.no_grad():
                check(lib.gsx_opacity_decay(ptr(op.data), ptr(vis_count), op.shape[0], 1, conf.opacity_decay,
                                            stream_ptr(op.device)), "gsx_opacity_decay")

    def optimize_map(self, window: List[Frame], n_iters: Optional[int] = None, regularize: bool = True,
                     early_stop: bool = True):
        """backend.py:249-362 without pruning/insertion."""
        n_iters = self.conf.num_iters_mapping if n_iters is None else n_iters
        stopper = StopOnPlateau(3, 0.012)
        last = None
        for _ in range(n_iters):
            total, photometric = self.step(window, regularize)
            last = (total, photometric)
            if early_stop:
                pm = photometric
                if self.shard.world_size > 1:
                    pm = self.shard.all_reduce_sum(pm * (len(self.shard.select(window)) / len(window)))
                if stopper.stop(pm.item()):
                    break
        outputs = self.last_outputs
        if outputs is not None:
            for f, d in zip(self.shard.select(window), outputs.depthmaps):
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
    csrc/window_opt.hip) and advanced by one single-wave launch at the end of the captured closure.  One refinement is
    ``max_eval + 1`` replays of one HIP graph and ONE read-back, instead of a `.item()` per closure (backend.py:501)
    and ~40 eager launches each.  The closure is the host version's: C <= 8 cameras rendered against the frozen map,
    photometric term only, pose-only backward.  Build one per window composition (the graph holds the addresses of the
    window's pose parameters and images); ``run()`` may be called repeatedly while the same keyframes are in it."""

    def __init__(self, splats: GaussianSplattingData, window: List[Frame], conf: Optional[MapConfig] = None,
                 max_eval: int = 25):
        from .rasterization import validate
        self.splats, self.window, self.conf, self.max_eval = splats, list(window), conf or MapConfig(), int(max_eval)
        self.params = [p for x in self.window if x.index != 0 for p in x.pose.parameters() if p.requires_grad]
        self.n = sum(p.numel() for p in self.params)
        if self.n == 0:
            raise ValueError("no learnable pose in the window")
        if self.n > 80 or len(self.params) > 16:
            raise ValueError("window too large for the device optimiser (80 parameters in 16 tensors)")
        dev = splats.means.device
        self.cameras = [x.camera for x in self.window]
        self.poses = [x.pose for x in self.window]
        self.gt_imgs = create_batch(self.window, lambda x: x.img)
        self.exposure = create_batch(self.window, lambda x: x.exposure_params).detach().clone()
        self._state = torch.zeros(int(lib.gsx_window_opt_state_bytes()), dtype=torch.uint8, device=dev)
        self._report = torch.zeros(8, dtype=torch.float32, device=dev)
        self._validate = validate
        self._sig = (int(splats.means.shape[0]), len(self.window), int(self.cameras[0].width), int(self.cameras[0].height))
        self.graph = None
        self.loss = None

    def _closure(self, advance: bool):
        from .losses import loss_and_grads
        for x in self.window:
            for p in x.pose.parameters():
                p.grad = None
        out = self.splats(self.cameras, self.poses, render_depth=False, need_n_touched=False)
        out2, v_render, _v_exp, _ = loss_and_grads(out._render, out.alphas, self.gt_imgs, self.exposure, None, None, -1,
                                                   -1 if out._betas_index is None else out._betas_index, 1.0, 0.0, 0.0,
                                                   0.0, 0 if self.conf.active_gs else 1)
        torch.autograd.backward([out._render], [v_render])
        loss = out2[0:1]
        if advance:
            import ctypes as C
            n = len(self.params)
            check(lib.gsx_window_opt_advance(
                self._state.data_ptr(), n, (C.c_void_p * n)(*[p.data_ptr() for p in self.params]),
                (C.c_void_p * n)(*[p.grad.data_ptr() for p in self.params]),
                (C.c_int * n)(*[p.numel() for p in self.params]), loss.data_ptr(), stream_ptr(loss.device)),
                "gsx_window_opt_advance")
        return loss

    def _frozen_map(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            flags = [p.requires_grad for p in self.splats.parameters()]
            for p in self.splats.parameters():
                p.requires_grad_(False)                       # only the poses move: pose-only projection backward
            try:
                yield
            finally:
                for p, r in zip(self.splats.parameters(), flags):
                    p.requires_grad_(r)
        return cm()

    def capture(self):
        from ._sync import capture_lock
        with capture_lock, self._frozen_map():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _attempt in range(4):
                    for _ in range(2):
                        self._closure(False)
                    if self._validate(signature=self._sig):
                        break
                else:
                    raise RuntimeError("intersection capacity keeps changing during warm-up")
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                self.loss = self._closure(True)

    def run_async(self):
        """one refinement without any read-back (the caller polls gslam_amd.rasterization.validate later): the
        graph must have been captured by an earlier run()"""
        assert self.graph is not None
        dev = self._state.device
        check(lib.gsx_window_opt_init(self._state.data_ptr(), self.n, 0, 0.0, 1.0, 10, 20, self.max_eval, 1e-7, 1e-7,
                                      stream_ptr(dev)), "gsx_window_opt_init")
        for _ in range(self.max_eval + 1):
            self.graph.replay()

    def run(self):
        """-> (last closure loss, number of closure evaluations), poses updated in place"""
        with torch.no_grad():
            self.exposure.copy_(create_batch(self.window, lambda x: x.exposure_params).detach())
        for attempt in range(2):
            saved = [p.detach().clone() for p in self.params]
            if self.graph is None:
                self.capture()
                with torch.no_grad():
                    for p, s0 in zip(self.params, saved):
                        p.copy_(s0)                           # the warm-up closures did not move them, the capture neither
            dev = self._state.device
            # torch.optim.LBFGS defaults of the reference call: lr 1, max_iter 20, max_eval 25, tolerance_grad 1e-7
            check(lib.gsx_window_opt_init(self._state.data_ptr(), self.n, 0, 0.0, 1.0, 10, 20, self.max_eval, 1e-7, 1e-7,
                                          stream_ptr(dev)), "gsx_window_opt_init")
            for _ in range(self.max_eval + 1):
                self.graph.replay()
            check(lib.gsx_window_opt_report(self._state.data_ptr(), self._report.data_ptr(), stream_ptr(dev)),
                  "gsx_window_opt_report")
            rep = self._report.cpu()
            if self._validate(signature=self._sig):
                return float(rep[4]), int(rep[1])
            with torch.no_grad():                             # tile lists outgrew the captured capacity: redo
                for p, s0 in zip(self.params, saved):
                    p.copy_(s0)
            self.graph = None
        raise RuntimeError("intersection buffers kept overflowing")


class GraphedBundleAdjuster:
    """A BA iteration over a FIXED window captured into HIP graphs and replayed: the ~45 launches of a step (pose
    chain, K1, binning, sort, K8, SSIM, loss, K9, K2, Adam, decay) cost one graph launch on the host.  Multi-GPU: two
    graphs (render+loss+backward | isotropic+Adam+decay) around the one eager all-reduce.  Needs a BundleAdjuster
    built with capturable=True; the window's tensors (images, poses, exposure) are updated in place between replays.
    ``validate()`` (gslam_amd.rasterization) must be polled by the caller: the intersection capacity is baked in."""

    def __init__(self, ba: BundleAdjuster, window: List[Frame], warmup: int = 3, regularize: bool = True):
        assert ba.optimizers.capturable, "build the BundleAdjuster with capturable=True"
        from .rasterization import validate
        self.ba, self.window = ba, window
        self.multi = ba.shard.world_size > 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 2)):
                ba.step(window, regularize)
            if not validate():                      # capacity grew: one more eager pass with the final capacity
                ba.step(window, regularize)
                assert validate()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # warm-up and capture share one side stream (autograd pins AccumulateGrad nodes to the stream of first use)
        self.graph = torch.cuda.CUDAGraph()
        self.graph2 = None
        # thread_local: the RCCL watchdog thread may query events while this thread captures
        mode = "thread_local" if self.multi else "global"
        if not self.multi:
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=mode):
                self.total, self.photometric = ba.step(window, regularize)
        else:
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=mode):
                self.total, self.photometric = ba.render_backward(window, regularize)
            ba.reduce()
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph2, pool=self.graph.pool(), stream=side, capture_error_mode=mode):
                ba.update()

    def step(self):
        self.graph.replay()
        if self.multi:
            self.ba.reduce()
            self.graph2.replay()
        self.ba.total_step += 1
        return self.total, self.photometric
