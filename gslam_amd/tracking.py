"""Per-frame camera tracking: ``Frontend.tracking_loss`` / ``igs_track_lbfgs`` / ``warp_track`` of
gslam/frontend.py:113-138, 604-662, 521-569 without the process / logging shell.  C = 1: tracking does not shard
("replicas only") - spare GPUs run bundle adjustment while one tracks."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import os

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
    params = list(new_frame.pose.parameters())
    if conf.learn_exposure_params:
        params.append(new_frame.exposure_params)
    opt = torch.optim.SGD(params, lr=conf.pose_optim_lr, momentum=0.8, nesterov=True)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=conf.pose_optim_lr_decay)
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
    """The tracking closure (C = 1 render forward + active-nerf loss + backward to the pose delta and the exposure
    parameters, gslam/frontend.py:621-649) captured once into a HIP graph over a persistent "slot" and replayed for
    every closure of every frame: ~12 kernel launches become one graph launch.

    ``device_optimizer=True`` (default) also moves the optimiser of frontend.py:613-658 - 10 Adam steps, then one
    strong-Wolfe L-BFGS step - onto the device (csrc/track_opt.h): the state machine is advanced by one single-wavefront
    kernel at the end of the captured closure, so a tracked frame is ``n_adam + max_eval + 1`` graph launches and ONE
    read-back at the end instead of one ``loss.item()`` per closure (frontend.py:648).  With ``False`` the optimiser
    logic stays on the host exactly as in the reference (torch.optim.Adam / torch.optim.LBFGS)."""

    def __init__(self, splats: GaussianSplattingData, camera, conf: Optional[TrackingConfig] = None,
                 device_optimizer: bool = True, max_eval: int = 25):
        from .losses import tracking_loss_and_grads
        from .primitives import PoseZhou
        from .rasterization import validate
        self.conf = conf or TrackingConfig()
        self.splats, self.camera = splats, camera
        dev = splats.means.device
        H, W = camera.height, camera.width
        self.pose = PoseZhou(torch.eye(4, device=dev)).to(dev)
        self.exposure = torch.zeros(2, device=dev, requires_grad=True)
        self.img = torch.zeros(H, W, 3, device=dev)
        self.params = [self.pose.dt, self.pose.dR, self.exposure]
        self._loss_fn = tracking_loss_and_grads
        self.graph = None
        self.loss = None
        self._validate = validate
        self.device_optimizer = device_optimizer
        self.max_eval = max_eval
        self._state = self._report = None
        self.fused_tail = bool(device_optimizer) and os.environ.get("GSX_TRACK_TAIL", "fused") != "split"
        if device_optimizer:
            from ._lib import lib
            self._state = torch.zeros(int(lib.gsx_track_opt_state_bytes()), dtype=torch.uint8, device=dev)
            self._report = torch.zeros(8, dtype=torch.float32, device=dev)
            # the view matrix the fused closure renders with: written by the closure's own tail for the next evaluation
            self._viewmat = torch.zeros(1, 4, 4, device=dev, requires_grad=True)
            self._viewmat._gsx_partials_only = True

    def _closure_body(self, advance: bool = False):
        if advance and self.fused_tail:
            return self._closure_fused()
        for p in self.params:
            p.grad = None               # AccumulateGrad then adopts the fresh gradient tensor (no accumulate kernel)
        # the reference renders the depth channel in its tracking closure too (frontend.py:627-631) but only reads it
        # under use_gt_depths (frontend.py:134-137), which this tracker does not implement: four channels instead of five
        out = self.splats([self.camera], [self.pose], render_depth=False, need_n_touched=False)  # n_touched: never read
        # value and analytic gradient in one pass (csrc/loss.hip); the backward is seeded at the render tensor
        out2, v_render, v_exposure = self._loss_fn(out, self.img, self.exposure)
        torch.autograd.backward([out._render], [v_render])
        self.exposure.grad = v_exposure
        loss = out2[0:1]
        if advance:
            import ctypes as C
            from ._lib import check, lib, stream_ptr
            n = len(self.params)
            check(lib.gsx_track_opt_advance(
                self._state.data_ptr(), n, (C.c_void_p * n)(*[p.data_ptr() for p in self.params]),
                (C.c_void_p * n)(*[p.grad.data_ptr() for p in self.params]),
                (C.c_int * n)(*[p.numel() for p in self.params]), loss.data_ptr(),
                stream_ptr(loss.device)), "gsx_track_opt_advance")
        return loss

    def _closure_fused(self):
        """the captured closure of the device optimiser: renders with the persistent view matrix, and ONE launch takes
        the projection backward's pose partials through the PoseZhou backward, the optimiser step and the PoseZhou
        forward of the new parameters (csrc/track_opt_impl.inc: track_opt_tail_kernel) - no autograd node and no
        launch of its own for the pose on either side of the render"""
        from ._lib import check, lib, stream_ptr
        from .ops import workspace
        out = self.splats._render([self.camera], self._viewmat, 'RGB', 0.5, need_n_touched=False)
        (loss_ws, n_rows, coef), v_render, _ = self._loss_fn(out, self.img, self.exposure, finish=False)
        torch.autograd.backward([out._render], [v_render])      # ends with the pose partials in the workspace
        N = int(self.splats.means.shape[0])
        dev = v_render.device
        ws = workspace(lib.gsx_project_bwd_workspace_bytes(N, 1), dev, "proj_bwd")
        check(lib.gsx_track_opt_tail(self._state.data_ptr(), ws.data_ptr(), int(lib.gsx_project_bwd_blocks(N)),
                                     self.pose.Rt.data_ptr(), self.pose.dt.data_ptr(), self.pose.dR.data_ptr(),
                                     self.exposure.data_ptr(), None, None, self._viewmat.data_ptr(),
                                     loss_ws.data_ptr(), n_rows, coef, stream_ptr(dev)), "gsx_track_opt_tail")
        return None

    def load(self, frame: Frame, prev_exposure: Optional[torch.Tensor] = None):
        with torch.no_grad():
            self.pose.Rt.copy_(frame.pose())
            self.pose.dR.zero_()
            self.pose.dt.zero_()
            self.img.copy_(frame.img)
            self.exposure.copy_(frame.exposure_params if prev_exposure is None else prev_exposure)
            if self.device_optimizer:
                self._viewmat.copy_(self.pose.Rt[None])               # dR = dt = 0: the first evaluation's view matrix

    def capture(self):
        from ._sync import capture_lock
        with capture_lock:
            self._capture()

    def _capture(self):
        # warm-up and capture run on the SAME side stream: autograd pins each leaf's AccumulateGrad node to the stream
        # it was first used on, and a capture on another stream would push the accumulation out of the graph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        sig = (int(self.splats.means.shape[0]), 1, int(self.camera.width), int(self.camera.height))
        with torch.cuda.stream(side):
            for attempt in range(4):                    # a warm-up render that overflowed has grown the capacity
                for _ in range(2):
                    self._closure_body()
                if self._validate(signature=sig):
                    break
            else:
                raise RuntimeError("intersection capacity keeps changing during warm-up")
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a backend thread of the same process may be allocating / synchronising on its own stream
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.loss = self._closure_body(advance=self.device_optimizer)

    def closure(self):
        if self.graph is None:
            return self._closure_body()
        self.graph.replay()
        return self.loss

    def _write_back(self, frame: Frame):
        with torch.no_grad():
            new_pose = self.pose().detach().clone()
            frame.pose.Rt.copy_(new_pose)
            frame.pose.dR.zero_()
            frame.pose.dt.zero_()
            frame.exposure_params.data.copy_(self.exposure.detach())

    def track(self, frame: Frame, prev_exposure: Optional[torch.Tensor] = None, max_eval: Optional[int] = None,
              sync: bool = True):
        """igs_track_lbfgs on the slot; writes the optimised pose / exposure back into ``frame``.  Returns
        (last_loss, n_closures); with the device optimiser and ``sync=False`` both are left on the device (the
        8-float report tensor of gsx_track_opt_report is returned instead) and the call does not block."""
        conf = self.conf
        self.load(frame, prev_exposure)
        if self.graph is None:
            self.capture()
            self.load(frame, prev_exposure)
        if self.device_optimizer:
            from ._lib import check, lib, stream_ptr
            dev = self.img.device
            me = self.max_eval if max_eval is None else max_eval
            n_par = sum(p.numel() for p in self.params)
            # torch.optim.LBFGS defaults of the reference call (frontend.py:613-619): max_iter 20, tolerance_grad 1e-7
            check(lib.gsx_track_opt_init(self._state.data_ptr(), n_par, conf.n_adam_warmup, conf.pose_optim_lr,
                                         conf.pose_optim_lr, conf.lbfgs_history, 20, me, 1e-7, 1e-9, stream_ptr(dev)),
                  "gsx_track_opt_init")
            for _ in range(conf.n_adam_warmup + me + 1):   # a line search may overshoot max_eval by one evaluation
                self.graph.replay()
            check(lib.gsx_track_opt_report(self._state.data_ptr(), self._report.data_ptr(), stream_ptr(dev)),
                  "gsx_track_opt_report")
            if not sync:
                self._write_back(frame)
                return self._report
            rep = self._report.cpu()
            # the map under the captured closure changes between frames (BA updates, in-place SYNC): if its tile lists
            # outgrew the capacity baked into the graph, re-capture with the grown buffers and track this frame again
            sig = (int(self.splats.means.shape[0]), 1, int(self.camera.width), int(self.camera.height))
            if not self._validate(signature=sig) and not getattr(self, "_retrying", False):
                self.graph = None
                self._retrying = True
                try:
                    return self.track(frame, prev_exposure, max_eval, sync)
                finally:
                    self._retrying = False
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
        self._write_back(frame)
        return last, n_evals
