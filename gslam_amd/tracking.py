"""Per-frame camera tracking: ``Frontend.tracking_loss`` / ``igs_track_lbfgs`` / ``warp_track`` of
gslam/frontend.py:113-138, 604-662, 521-569 without the process / logging shell.  C = 1: tracking does not shard
("replicas only") - spare GPUs run bundle adjustment while one tracks."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .map import GaussianSplattingData
from .primitives import Frame
from .warp import Warp


@dataclass
class TrackingConfig:
    """Subset of gslam/frontend.py:44-61 (same names / defaults)."""
    device: str = 'cuda'
    num_tracking_iters: int = 200          # warp tracker iterations (frontend.py:47-51)
    photometric_loss: str = 'active-nerf'
    pose_optim_lr: float = 0.002
    pose_optim_lr_decay: float = 0.99
    learn_exposure_params: bool = True
    use_gt_depths: bool = False
    n_adam_warmup: int = 10                # frontend.py:651
    lbfgs_history: int = 5                 # frontend.py:613-619


def tracking_loss(conf: TrackingConfig, gt_img, rendered_img, betas=None, rendered_depth=None, gt_depth=None):
    """gslam/frontend.py:113-138."""
    error = rendered_img - gt_img
    if conf.photometric_loss == 'l1':
        loss = error.abs().mean()
    elif conf.photometric_loss == 'mse':
        loss = error.square().mean()
    elif conf.photometric_loss == 'active-nerf':
        loss = (error.square().sum(dim=-1) * betas.pow(-2.0)).mean()
    elif conf.photometric_loss == 'none':
        loss = error
    else:
        raise ValueError(conf.photometric_loss)
    if conf.use_gt_depths:
        depth_error = (rendered_depth - gt_depth)[gt_depth > 0.0]
        loss = loss + depth_error.abs().mean() * 0.01
    return loss


def igs_track_lbfgs(splats: GaussianSplattingData, new_frame: Frame, conf: Optional[TrackingConfig] = None,
                    prev_exposure: Optional[torch.Tensor] = None, max_eval: Optional[int] = None):
    """Default tracker (gslam/frontend.py:604-662): 10 Adam steps then one L-BFGS(strong Wolfe) step on the pose
    delta (+ exposure).  Every closure is one C=1 render forward+backward.  Returns (last_loss, n_closures)."""
    conf = conf or TrackingConfig()
    n_evals = 0
    params = list(new_frame.pose.parameters())
    if conf.learn_exposure_params:
        if prev_exposure is not None:
            new_frame.exposure_params.data = prev_exposure.clone().detach()
        params.append(new_frame.exposure_params)
    kw = {} if max_eval is None else {"max_eval": max_eval}
    optimizer = torch.optim.LBFGS(params, history_size=conf.lbfgs_history, line_search_fn='strong_wolfe',
                                  tolerance_change=1e-9, lr=conf.pose_optim_lr, **kw)
    last_loss = None

    def closure():
        nonlocal n_evals, last_loss
        n_evals += 1
        if torch.is_grad_enabled():
            optimizer.zero_grad()
        outputs = splats([new_frame.camera], [new_frame.pose], render_depth=True)
        rendered_rgb = outputs.rgbs[0]
        if conf.learn_exposure_params:
            rendered_rgb = rendered_rgb * new_frame.exposure_params[0].exp() + new_frame.exposure_params[1]
        loss = tracking_loss(conf, rendered_rgb, new_frame.img, outputs.betas[0], outputs.depthmaps[0],
                             new_frame.gt_depth)
        if loss.requires_grad:
            loss.backward()
        last_loss = loss.item()
        return loss

    warm = torch.optim.Adam(params, conf.pose_optim_lr)
    for _ in range(conf.n_adam_warmup):
        closure()
        warm.step()
        warm.zero_grad()
    optimizer.step(closure)
    return last_loss, n_evals


def warp_track(new_frame: Frame, ref_frame: Frame, ref_img: torch.Tensor, ref_depth: torch.Tensor,
               conf: Optional[TrackingConfig] = None, n_iters: Optional[int] = None):
    """Alternative tracker (gslam/frontend.py:521-569): SGD-Nesterov on the pose through the depth-based warp of the
    last keyframe; masked L1 with exposure affine."""
    conf = conf or TrackingConfig()
    n_iters = conf.num_tracking_iters if n_iters is None else n_iters
    K = new_frame.camera.intrinsics
    warp = Warp(K, new_frame.camera.height, new_frame.camera.width).to(K.device)
    # frontend.py:193-210: Nesterov SGD on the pose, exponential decay, the exposure pair as a second group at lr 0.01
    opt = torch.optim.SGD(list(new_frame.pose.parameters()), lr=conf.pose_optim_lr, momentum=0.8, nesterov=True)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=conf.pose_optim_lr_decay)
    if conf.learn_exposure_params:
        opt.add_param_group({'params': new_frame.exposure_params, 'lr': 0.01})
    loss = None
    ref_pose = ref_frame.pose().detach()
    for _ in range(n_iters):
        opt.zero_grad()
        result, _, keep = warp(ref_pose, new_frame.pose(), ref_img, ref_depth)
        if conf.learn_exposure_params:
            result = result * new_frame.exposure_params[0].exp() + new_frame.exposure_params[1]
        loss = ((result - new_frame.img).abs() * keep[..., None]).sum() / keep.sum().clamp(min=1) / 3.0
        loss.backward()
        opt.step()
        sched.step()
    return loss


class GraphedTracker:
    """``igs_track_lbfgs`` (gslam/frontend.py:604-662) on a launch plan: the tracking closure - C = 1 render forward,
    active-nerf loss, backward to the pose delta and the exposure parameters (:621-649) - is a fixed chain of libgsx
    launches over a persistent slot (gslam_amd.plan.TrackClosure), recorded once into a HIP graph and replayed for every
    closure of every frame.  No autograd graph and no allocation inside the closure.

    ``device_optimizer=True`` (default) also moves the optimiser of frontend.py:613-658 - 10 Adam steps, then one
    strong-Wolfe L-BFGS step - onto the device (csrc/track_opt.h): the state machine is advanced at the end of the captured
    closure, so a tracked frame is ``n_adam + max_eval + 1`` graph launches and ONE read-back at the end instead of one
    ``loss.item()`` per closure (frontend.py:648).  With ``False`` the optimiser logic stays on the host exactly as in the
    reference (torch.optim.Adam / torch.optim.LBFGS) and only the closure is a graph.

    ``tail``: 'fused' (default with the device optimiser: pose backward, loss finish, optimiser step and next view matrix
    in one launch) or 'split' (the same as separate launches; kept as an independent implementation for the tests)."""

    def __init__(self, splats: GaussianSplattingData, camera, conf: Optional[TrackingConfig] = None,
                 device_optimizer: bool = True, max_eval: int = 25, tail: Optional[str] = None):
        from .plan import TrackClosure
        from .primitives import PoseZhou
        self.conf = conf or TrackingConfig()
        self.splats, self.camera = splats, camera
        self.device_optimizer = bool(device_optimizer)
        self.max_eval = int(max_eval)
        if tail is None:
            tail = 'fused' if device_optimizer else 'host'
        if not device_optimizer:
            tail = 'host'
        self.fused_tail = tail == 'fused'
        self.plan = TrackClosure(splats, camera, tail=tail)
        c = self.plan
        dev = c.dev
        # the slot seen as the reference's objects: a PoseZhou whose Rt / dR / dt ARE the plan's buffers, the exposure pair
        self.pose = PoseZhou(torch.eye(4, device=dev)).to(dev)
        self.pose.Rt = c.Rt
        self.pose.dt = torch.nn.Parameter(c.dt)
        self.pose.dR = torch.nn.Parameter(c.dR)
        self.exposure = c.exposure.requires_grad_(True)
        self.img = c.img
        self.params = [self.pose.dt, self.pose.dR, self.exposure]
        self._grads = [c.g_dt, c.g_dR, c.g_exposure]
        self._state, self._report = c.state, c.report
        self.loss = c.loss
        self._retrying = 0

    @property
    def graph(self):
        return self.plan.graph if self.plan.graph.captured else None

    def matches(self, splats) -> bool:
        return self.plan.r.matches(splats)

    def load(self, frame: Frame, prev_exposure: Optional[torch.Tensor] = None):
        exposure = frame.exposure_params if prev_exposure is None else prev_exposure
        pose = frame.pose
        dev = self.plan.dev

        def raw_ok(t, shape=None):              # what the row kernels read through a raw pointer
            return (torch.is_tensor(t) and t.device == dev and t.dtype == torch.float32 and t.is_contiguous()
                    and (shape is None or tuple(t.shape) == shape))

        if (all(hasattr(pose, k) for k in ("Rt", "dR", "dt")) and raw_ok(pose.Rt, (4, 4)) and raw_ok(pose.dR)
                and raw_ok(pose.dt) and raw_ok(frame.img, tuple(self.img.shape[-3:])) and raw_ok(exposure)):
            self.plan.load_frame(pose, frame.img, exposure)         # three launches, no torch PoseZhou forward
        else:                                                       # anything else is converted by copy_ (dtype, layout, device)
            self.plan.load(frame.pose().detach(), frame.img, exposure.detach())

    def capture(self):
        """probe the tile-list capacity for the loaded frame, warm up and record the closure (the slot is left as loaded)"""
        from ._sync import capture_lock
        with capture_lock:
            self.plan.prepare()

    def closure(self):
        """one evaluation at the slot's current parameters: loss [1] on the device; with the host optimiser the
        gradients are attached to ``self.params``"""
        if self.plan.graph.captured:
            self.plan.graph.replay()
        else:
            if self.plan.r.capacity == 0:
                self.plan.r.probe()
            from .plan import current_stream_ptr
            self.plan.enqueue(current_stream_ptr(self.plan.dev))
        if not self.device_optimizer:
            for p, g in zip(self.params, self._grads):
                p.grad = g
        return self.loss

    def _write_back(self, frame: Frame):
        pose = frame.pose
        if self.fused_tail and all(hasattr(pose, k) for k in ("Rt", "dR", "dt")) and pose.Rt.is_cuda \
                and pose.Rt.is_contiguous() and frame.exposure_params.is_contiguous():
            with torch.no_grad():
                self.plan.store_frame(pose, frame.exposure_params)  # one launch
            return
        with torch.no_grad():
            # fused tail: the closure's last launch left the view matrix of the final parameters in the plan
            new_pose = self.plan.r.viewmats[0] if self.fused_tail else self.pose().detach().clone()
            frame.pose.Rt.copy_(new_pose)
            frame.pose.dR.zero_()
            frame.pose.dt.zero_()
            frame.exposure_params.data.copy_(self.exposure.detach())

    def track(self, frame: Frame, prev_exposure: Optional[torch.Tensor] = None, max_eval: Optional[int] = None,
              sync: bool = True):
        """igs_track_lbfgs on the slot; writes the optimised pose / exposure back into ``frame``.  Returns
        (last_loss, n_closures); with the device optimiser and ``sync=False`` both are left on the device (the
        8-float report tensor of gsx_track_opt_report is returned instead) and the call does not block - the caller
        then polls ``capacity_ok()`` now and then."""
        conf = self.conf
        self.load(frame, prev_exposure)
        if not self.plan.graph.captured or self.plan.r.stale:
            self.capture()
            self.load(frame, prev_exposure)
        if self.device_optimizer:
            me = self.max_eval if max_eval is None else int(max_eval)
            self.plan.init_optimizer(conf.n_adam_warmup, conf.pose_optim_lr, conf.lbfgs_history, me)
            # a line search may overshoot max_eval by one evaluation
            self.plan.launch(conf.n_adam_warmup + me + 1)
            rep = self.plan.read_report()
            if not sync:
                self._write_back(frame)
                return rep
            rep = rep.cpu()
            # the map under the captured closure changes between frames (BA updates, in-place SYNC): if its tile lists
            # outgrew the capacity baked into the graph, re-capture with the grown buffers and track this frame again
            # (frame.pose is only written after a clean run: every attempt starts from the frame's own pose and exposure)
            if not self.plan.r.check_capacity():
                return self._retry(frame, prev_exposure, max_eval, sync)
            self._write_back(frame)
            return float(rep[4]), int(rep[1])
        n_evals = 0
        last = None

        def closure():
            nonlocal n_evals, last
            n_evals += 1
            loss = self.closure()
            last = loss.item()                      # frontend.py:648 ("this sync is okay because lbfgs syncs anyway")
            return loss

        warm = torch.optim.Adam(self.params, conf.pose_optim_lr)
        for _ in range(conf.n_adam_warmup):
            closure()
            warm.step()
        kw = {} if max_eval is None else {"max_eval": max_eval}
        lbfgs = torch.optim.LBFGS(self.params, history_size=conf.lbfgs_history, line_search_fn='strong_wolfe',
                                  tolerance_change=1e-9, lr=conf.pose_optim_lr, **kw)
        lbfgs.step(closure)
        if not self.plan.r.check_capacity():
            return self._retry(frame, prev_exposure, max_eval, sync)
        self._write_back(frame)
        return last, n_evals

    MAX_ATTEMPTS = 3

    def _retry(self, frame, prev_exposure, max_eval, sync):
        """the closure's tile lists overflowed during this frame (the buffers have been grown, the plan is stale): track the
        frame again from its own pose - nothing of the truncated attempt was written back - and give up loudly after
        MAX_ATTEMPTS instead of handing out a pose optimised on truncated renders"""
        self._retrying += 1
        try:
            if self._retrying >= self.MAX_ATTEMPTS:
                raise RuntimeError(f"tracking closure: tile lists overflowed in {self.MAX_ATTEMPTS} consecutive attempts "
                                   f"(capacity {self.plan.r.capacity}, M {self.plan.r.last_M})")
            return self.track(frame, prev_exposure, max_eval, sync)
        finally:
            self._retrying -= 1

    def capacity_ok(self) -> bool:
        """blocking check of the sticky overflow status (for callers of ``track(sync=False)``)"""
        return self.plan.r.check_capacity()
