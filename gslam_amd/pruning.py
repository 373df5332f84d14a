"""Map pruning between bundle-adjustment iterations: gslam/pruning.py (same names, arguments and return values).

``prune_using_mask`` re-packs the seven per-Gaussian parameters, the two Adam moments of each optimised one and any
extra per-Gaussian arrays in ONE launch (csrc/maintain.hip) instead of one boolean-indexing kernel per array, and
rebuilds the optimiser bookkeeping exactly like the reference (which follows gsplat's strategy ops).  ``optimizers``
may be the reference's ``Dict[str, Optimizer]`` (one optimiser per parameter name, backend.py:565-602) or this
package's ``MapOptimizers`` (one multi-tensor FusedAdam).  The strategies only build masks; they are the reference's
formulas verbatim."""
from __future__ import annotations

import ctypes as C
from abc import ABC
from typing import Dict, List, Optional, Sequence, Union

import torch

from ._lib import check, lib, stream_ptr
from .map import GaussianSplattingData


def _row_words(t: torch.Tensor) -> int:
    if t.element_size() not in (4, 8) or not t.is_cuda:
        raise TypeError("per-Gaussian arrays must be 4- or 8-byte GPU tensors (no CPU fallback)")
    per_row = (t.numel() // t.shape[0]) if t.shape[0] > 0 else int(torch.Size(t.shape[1:]).numel())
    return per_row * (t.element_size() // 4)


def gather_rows(tensors: Sequence[torch.Tensor], index: torch.Tensor) -> List[torch.Tensor]:
    """[t[index] for t in tensors] in one launch (index: int64 device tensor of source rows)."""
    index = index.to(torch.int64).contiguous()
    n_out = int(index.shape[0])
    srcs = [t.detach().contiguous() for t in tensors]
    outs = [torch.empty((n_out,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in srcs]
    if n_out == 0 or not srcs:
        return outs
    n = len(srcs)
    for i in range(0, n, 32):
        part_s, part_o = srcs[i:i + 32], outs[i:i + 32]
        m = len(part_s)
        check(lib.gsx_gather_rows(m, (C.c_void_p * m)(*[t.data_ptr() for t in part_s]),
                                  (C.c_void_p * m)(*[t.data_ptr() for t in part_o]),
                                  (C.c_int * m)(*[_row_words(t) for t in part_s]), index.data_ptr(), n_out,
                                  int(part_s[0].shape[0]), stream_ptr(index.device)), "gsx_gather_rows")
    return outs


def concat_rows(a: Sequence[torch.Tensor], b: Sequence[Optional[torch.Tensor]], n_b: int) -> List[torch.Tensor]:
    """[cat(a_k, b_k) for k] in one launch; b_k None = n_b zero rows."""
    srcs = [t.detach().contiguous() for t in a]
    n_a = int(srcs[0].shape[0])
    bs = [None if t is None else t.detach().to(s.dtype).contiguous() for t, s in zip(b, srcs)]
    outs = [torch.empty((n_a + n_b,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in srcs]
    if n_a + n_b == 0:
        return outs
    for i in range(0, len(srcs), 32):
        ps, pb, po = srcs[i:i + 32], bs[i:i + 32], outs[i:i + 32]
        m = len(ps)
        check(lib.gsx_concat_rows(m, (C.c_void_p * m)(*[t.data_ptr() for t in ps]), n_a,
                                  (C.c_void_p * m)(*[None if t is None else t.data_ptr() for t in pb]), n_b,
                                  (C.c_void_p * m)(*[t.data_ptr() for t in po]),
                                  (C.c_int * m)(*[_row_words(t) for t in po]), stream_ptr(po[0].device)),
              "gsx_concat_rows")
    return outs


def _optimizer_of(optimizers, name: str):
    """(optimizer, group index) that owns splat parameter ``name`` or (None, None)."""
    if optimizers is None:
        return None, None
    if isinstance(optimizers, dict):
        opt = optimizers.get(name)
        return (opt, 0) if opt is not None else (None, None)
    opt = getattr(optimizers, "splat_opt", None)          # gslam_amd.mapping.MapOptimizers
    if opt is None:
        return None, None
    from .mapping import SPLAT_LRS
    for gi, (pname, _lr) in enumerate(SPLAT_LRS):
        if pname == name:
            return opt, gi
    return None, None


def _state_tensors(opt, param) -> Dict[str, torch.Tensor]:
    st = opt.state.get(param, {})
    return {k: v for k, v in st.items() if k != 'step' and torch.is_tensor(v) and v.dim() >= 1
            and v.shape[0] == param.shape[0]}


def _rebuild(splats: GaussianSplattingData, optimizers, new_values: Dict[str, torch.Tensor],
             new_states: Dict[str, Dict[str, torch.Tensor]]):
    """swap the parameters (and their optimiser state) for the re-packed arrays: gslam/pruning.py:24-47"""
    for name, old in list(splats.named_parameters()):
        new_p = torch.nn.Parameter(new_values[name], requires_grad=old.requires_grad)
        opt, gi = _optimizer_of(optimizers, name)
        if opt is not None:
            state = opt.state.pop(old, {})
            for key, val in new_states.get(name, {}).items():
                state[key] = val
            opt.state[new_p] = state
            groups = [gi] if not isinstance(optimizers, dict) else range(len(opt.param_groups))
            for g in groups:
                opt.param_groups[g]['params'] = [new_p]
        splats.__setattr__(name, new_p)


@torch.no_grad()
def prune_using_mask(splats: GaussianSplattingData, optimizers, keep_mask: torch.Tensor,
                     per_gaussian_params: Optional[List[torch.Tensor]] = None):
    """gslam/pruning.py:10-55.  Returns the number of pruned Gaussians (0 when the mask would empty the map, :16-17)."""
    n_keep = int(keep_mask.sum().item())
    if n_keep == 0:
        return 0
    n_pruned = keep_mask.shape[0] - n_keep
    index = torch.nonzero(keep_mask, as_tuple=False).reshape(-1)
    names, tensors, slots = [], [], []
    for name, p in splats.named_parameters():
        tensors.append(p.data)
        slots.append(("param", name, None))
        opt, _gi = _optimizer_of(optimizers, name)
        if opt is not None:
            for key, val in _state_tensors(opt, p).items():
                tensors.append(val)
                slots.append(("state", name, key))
    n_extra = 0 if per_gaussian_params is None else len(per_gaussian_params)
    for i in range(n_extra):
        tensors.append(per_gaussian_params[i])
        slots.append(("extra", i, None))
    outs = gather_rows(tensors, index)
    new_values, new_states = {}, {}
    for (kind, a, b), t in zip(slots, outs):
        if kind == "param":
            new_values[a] = t
        elif kind == "state":
            new_states.setdefault(a, {})[b] = t
        else:
            per_gaussian_params[a] = t
    _rebuild(splats, optimizers, new_values, new_states)
    return n_pruned


class PruningStrategy(ABC):
    def step(self, splats: GaussianSplattingData, optimizers):
        return


class PruneLowOpacity(PruningStrategy):
    """gslam/pruning.py:63-76"""

    def __init__(self, min_opacity: float):
        self.min_opacity = min_opacity

    @torch.no_grad()
    def step(self, splats: GaussianSplattingData, optimizers=None) -> torch.Tensor:
        return torch.sigmoid(splats.opacities) < self.min_opacity


class PruneByVisibility(PruningStrategy):
    """gslam/pruning.py:79-103"""

    def __init__(self, window_size, min_visibility):
        self.window_size = window_size
        self.min_visibility = min_visibility

    @torch.no_grad()
    def step(self, splats: GaussianSplattingData, optimizers, visibility_counts: torch.Tensor, latest_kf_age: int):
        newly_added = splats.ages > (latest_kf_age - 3)          # monogs uses 3
        return newly_added & (visibility_counts < self.min_visibility)


class PruneLargeGaussians(PruningStrategy):
    """gslam/pruning.py:106-121: screen-space footprint above ``max_radius`` (radii: max over the rendered views)"""

    def __init__(self, max_radius: float):
        self.max_radius = max_radius

    @torch.no_grad()
    def step(self, splats: GaussianSplattingData, optimizers, radii: torch.Tensor):
        return radii > self.max_radius


class PruneIllConditionedGaussians(PruningStrategy):
    """gslam/pruning.py:124-139: visible (radius > 0) but touching no pixel in more than ``max_frames_thing`` views"""

    def __init__(self, max_frames_thing):
        self.max_frames_thing = max_frames_thing

    @torch.no_grad()
    def step(self, splats: GaussianSplattingData, optimizers, radii: torch.Tensor, n_touched: torch.Tensor):
        return ((radii > 0) & (n_touched == 0)).sum(dim=0) > self.max_frames_thing
