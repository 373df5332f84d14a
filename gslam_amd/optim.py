"""Multi-tensor fused Adam on libgsx.so: one launch updates up to 8 parameter tensors with per-tensor learning
rates.  Semantics = torch.optim.Adam(fused=True) with the defaults the reference uses (betas (0.9, 0.999), eps 1e-8,
no weight decay, no amsgrad): gslam/backend.py:565-602 creates six such optimisers (one per splat attribute) plus a
pose optimiser; this class folds them into one kernel reading p, g, m, v once (28 B per element)."""
from __future__ import annotations

import ctypes as C
from typing import Iterable

import torch

from ._lib import check, lib, stream_ptr

_MAX = 32          # tensors per launch (csrc/misc.hip ADAM_MAX)
_MAX_COUNTERS = 16  # device step counters bumped per gsx_counters_add launch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 capturable: bool = False):
        """capturable=True keeps the step counter on the device (one int64 per parameter, like torch's capturable
        Adam) so that ``step()`` can be captured into a HIP graph and replayed."""
        self.capturable = capturable
        self._shared_step = None          # device step shared by the groups given at construction
        self._constructing = True
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._constructing = False

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        g = self.param_groups[-1]
        g["_host_step"] = 0
        g["_step_dev"] = None
        g["_shared"] = self._constructing

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        step_all([self])
        return loss

    def _collect(self, buckets: dict, bumped: dict):
        """adds this optimiser's live parameters to ``buckets`` (keyed by the scalar state a launch shares) and its
        device step counters to ``bumped``"""
        for group in self.param_groups:
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            group["_host_step"] += 1
            step_dev = None
            if self.capturable:
                dev0 = live[0].device
                if group["_shared"]:
                    if self._shared_step is None:
                        self._shared_step = torch.full((1,), group["_host_step"] - 1, dtype=torch.int64, device=dev0)
                    step_dev = self._shared_step
                else:
                    if group["_step_dev"] is None:
                        group["_step_dev"] = torch.full((1,), group["_host_step"] - 1, dtype=torch.int64, device=dev0)
                    step_dev = group["_step_dev"]
                bumped.setdefault(id(step_dev), step_dev)
            for p in live:
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                    raise RuntimeError("FusedAdam needs contiguous float32 parameters on the GPU (no CPU fallback)")
                key = (group["betas"], group["eps"], -1 if self.capturable else group["_host_step"], p.device)
                buckets.setdefault(key, []).append((p, p.grad.contiguous(), st, float(group["lr"]), step_dev))


@torch.no_grad()
def step_all(optimizers, decay=None):
    """One update of several FusedAdam instances with as few launches as their scalar state allows: every device step
    counter of every optimiser is bumped by ONE gsx_counters_add, and parameters that share (betas, eps) go into the
    same multi-tensor launch whichever optimiser owns them (the splat and pose Adams of gslam/backend.py:554-602 become
    one launch).  decay = (param, mask_int32, min_count, factor): param *= factor where mask > min_count, applied by the
    launch that updates ``param`` (capturable optimisers only) - the opacity decay of backend.py:356-359."""
    buckets: dict = {}
    bumped: dict = {}
    for opt in optimizers:
        opt._collect(buckets, bumped)
    # (Letting the Adam launch advance the counters itself - a ticket per workgroup, the last one bumps - was measured
    # 8 % slower on the whole BA step: ~700 atomics on one address cost more than the 4.7 us launch they replace.)
    if bumped:                                         # all device step counters of this update in one tiny launch
        ctrs = list(bumped.values())
        for i in range(0, len(ctrs), _MAX_COUNTERS):
            part = ctrs[i:i + _MAX_COUNTERS]
            check(lib.gsx_counters_add(len(part), (C.c_void_p * len(part))(*[t.data_ptr() for t in part]), 1,
                                       stream_ptr(part[0].device)), "gsx_counters_add")
    decayed = False
    for (betas, eps, step, dev), items in buckets.items():
        for i in range(0, len(items), _MAX):
            chunk = items[i:i + _MAX]
            n = len(chunk)
            arr = lambda xs: (C.c_void_p * n)(*xs)
            common = (n, arr([c_[0].data_ptr() for c_ in chunk]), arr([c_[1].data_ptr() for c_ in chunk]),
                      arr([c_[2]["exp_avg"].data_ptr() for c_ in chunk]),
                      arr([c_[2]["exp_avg_sq"].data_ptr() for c_ in chunk]),
                      (C.c_int64 * n)(*[c_[0].numel() for c_ in chunk]),
                      (C.c_float * n)(*[c_[3] for c_ in chunk]), float(betas[0]), float(betas[1]), float(eps))
            if step == -1:
                k = -1
                if decay is not None:
                    k = next((j for j, c_ in enumerate(chunk) if c_[0] is decay[0]), -1)
                steps = arr([c_[4].data_ptr() for c_ in chunk])
                if k >= 0:
                    mask = decay[1]
                    if not (mask.dtype == torch.int32 and mask.is_contiguous() and mask.numel() == decay[0].numel()):
                        raise RuntimeError("decay mask must be a contiguous int32 tensor with one entry per element")
                    check(lib.gsx_adam_multi_steps_decay(*common, steps, k, mask.data_ptr(), int(decay[2]),
                                                         float(decay[3]), stream_ptr(dev)), "gsx_adam_multi_steps_decay")
                    decayed = True
                else:
                    check(lib.gsx_adam_multi_steps(*common, steps, stream_ptr(dev)), "gsx_adam_multi_steps")
            else:
                check(lib.gsx_adam_multi(*common, int(step), None, stream_ptr(dev)), "gsx_adam_multi")
    return decayed


class AdamPack:
    """The update of ``step_all`` with every argument resolved ahead of time, for launch plans (gslam_amd.plan): the
    gradient of each parameter is a caller-owned persistent buffer (``grad_of[id(p)]``) instead of ``p.grad``, the Adam
    moments and the device step counters of the (capturable) optimisers are created up front, and ``launch(stream)``
    issues one gsx_counters_add plus one multi-tensor launch per (betas, eps) class - nothing is allocated, so the call
    can be recorded into a HIP graph.  ``decay`` as in step_all.  Parameters without an entry in ``grad_of`` are skipped.
    ``gate``: a device float; the launches do nothing while gate[0] > 0 (gsx_adam_multi_steps_gated: no update from an
    iteration whose render overflowed its tile lists)."""

    def __init__(self, optimizers, grad_of: dict, decay=None, gate: torch.Tensor | None = None, pieces=None,
                 bump: bool = True):
        """pieces (sharded update, gslam_amd.dist.StepBucket): explicit slices instead of whole parameters - a list of
        dicts {p, g, m, v: 1-D float32 views of equal length, lr, betas, eps, step: device int64 [1], group: the owning
        param group}; ``decay[0]`` then names one of the ``p`` views and the mask has one int32 per element of it."""
        self._bump = bool(bump)
        self._gate = gate
        self._groups = []
        counters: dict = {}
        classes: dict = {}
        for pc in (pieces or []):
            p, g, m, v = pc["p"], pc["g"], pc["m"], pc["v"]
            if not all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == p.numel() for t in (p, g, m, v)):
                raise RuntimeError("AdamPack: a piece needs four contiguous float32 GPU views of equal length")
            counters.setdefault(id(pc["step"]), pc["step"])
            if pc.get("group") is not None and all(pc["group"] is not g_ for g_ in self._groups):
                self._groups.append(pc["group"])
            key = (tuple(pc["betas"]), float(pc["eps"]), p.device)
            classes.setdefault(key, []).append((p, g, {"exp_avg": m, "exp_avg_sq": v}, float(pc["lr"]), pc["step"]))
        for opt in optimizers:
            if opt is None:
                continue
            if not opt.capturable:
                raise RuntimeError("AdamPack needs FusedAdam(capturable=True): the step counters must live on the device")
            for group in opt.param_groups:
                live = [p for p in group["params"] if id(p) in grad_of]
                if not live:
                    continue
                dev0 = live[0].device
                if group["_shared"]:
                    if opt._shared_step is None:
                        opt._shared_step = torch.full((1,), group["_host_step"], dtype=torch.int64, device=dev0)
                    step_dev = opt._shared_step
                else:
                    if group["_step_dev"] is None:
                        group["_step_dev"] = torch.full((1,), group["_host_step"], dtype=torch.int64, device=dev0)
                    step_dev = group["_step_dev"]
                counters.setdefault(id(step_dev), step_dev)
                self._groups.append(group)
                for p in live:
                    g = grad_of[id(p)]
                    if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                        raise RuntimeError("FusedAdam needs contiguous float32 parameters on the GPU (no CPU fallback)")
                    if not (g.is_contiguous() and g.dtype == torch.float32 and g.numel() == p.numel()):
                        raise RuntimeError("AdamPack: gradient buffers must be contiguous float32 of the parameter's size")
                    st = opt.state[p]
                    if not st:
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    key = (tuple(group["betas"]), float(group["eps"]), p.device)
                    classes.setdefault(key, []).append((p, g, st, float(group["lr"]), step_dev))
        self._keep = (counters, classes)
        # bump = False: the step counters of these pieces are advanced by ANOTHER pack launched before this one in the same
        # iteration (one update cut into several launches: gslam_amd.plan.MappingStep's ranged exchange)
        ctrs = list(counters.values()) if bump else []
        self._counter_calls = []
        for i in range(0, len(ctrs), _MAX_COUNTERS):
            part = ctrs[i:i + _MAX_COUNTERS]
            self._counter_calls.append((len(part), (C.c_void_p * len(part))(*[t.data_ptr() for t in part])))
        self._adam_calls = []
        self.decay_applied = False
        for (betas, eps, _dev), items in classes.items():
            for i in range(0, len(items), _MAX):
                chunk = items[i:i + _MAX]
                n = len(chunk)
                arr = lambda xs: (C.c_void_p * n)(*xs)
                common = (n, arr([c_[0].data_ptr() for c_ in chunk]), arr([c_[1].data_ptr() for c_ in chunk]),
                          arr([c_[2]["exp_avg"].data_ptr() for c_ in chunk]),
                          arr([c_[2]["exp_avg_sq"].data_ptr() for c_ in chunk]),
                          (C.c_int64 * n)(*[c_[0].numel() for c_ in chunk]),
                          (C.c_float * n)(*[c_[3] for c_ in chunk]), float(betas[0]), float(betas[1]), float(eps))
                steps = arr([c_[4].data_ptr() for c_ in chunk])
                k = -1
                if decay is not None:
                    k = next((j for j, c_ in enumerate(chunk) if c_[0] is decay[0]), -1)
                if k >= 0:
                    mask = decay[1]
                    if not (mask.dtype == torch.int32 and mask.is_contiguous() and mask.numel() == decay[0].numel()):
                        raise RuntimeError("decay mask must be a contiguous int32 tensor with one entry per element")
                    self._adam_calls.append((common, steps, (k, mask.data_ptr(), int(decay[2]), float(decay[3]))))
                    self._keep = self._keep + (mask,)
                    self.decay_applied = True
                else:
                    self._adam_calls.append((common, steps, None))

    def launch(self, stream_ptr_: int):
        if self._gate is not None:
            gate = self._gate.data_ptr()
            for n, ptrs in self._counter_calls:
                check(lib.gsx_counters_add_gated(n, ptrs, 1, gate, stream_ptr_), "gsx_counters_add_gated")
            for common, steps, dec in self._adam_calls:
                d = dec if dec is not None else (-1, None, 0, 1.0)
                check(lib.gsx_adam_multi_steps_gated(*common, steps, d[0], d[1], d[2], d[3], gate, stream_ptr_),
                      "gsx_adam_multi_steps_gated")
            return
        for n, ptrs in self._counter_calls:
            check(lib.gsx_counters_add(n, ptrs, 1, stream_ptr_), "gsx_counters_add")
        for common, steps, dec in self._adam_calls:
            if dec is not None:
                check(lib.gsx_adam_multi_steps_decay(*common, steps, dec[0], dec[1], dec[2], dec[3], stream_ptr_),
                      "gsx_adam_multi_steps_decay")
            else:
                check(lib.gsx_adam_multi_steps(*common, steps, stream_ptr_), "gsx_adam_multi_steps")

    def note_steps(self, n: int = 1):
        """host-side bookkeeping for ``n`` issued updates (the counters that matter are on the device)"""
        for g in self._groups:
            g["_host_step"] += n
