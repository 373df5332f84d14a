"""PoseZhou / Camera / Frame of gslam/primitives.py (the parts on the hot path).

PoseZhou = fixed ``Rt`` times a learnable delta (6D rotation by Gram-Schmidt + translation), 9 scalars per pose;
this is the SE(3) parametrisation the v_viewmats gradient of K2 flows into (primitives.py:15-36,40-92)."""
from __future__ import annotations

from copy import deepcopy
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


class PoseZhou(torch.nn.Module):
    def __init__(self, _pose: Optional[torch.Tensor] = None, is_learnable: bool = True):
        super().__init__()
        self.is_learnable = is_learnable
        self.register_buffer("Rt", torch.eye(4) if _pose is None else _pose)
        self.dt = torch.nn.Parameter(torch.zeros(3, dtype=torch.float32, device=self.Rt.device),
                                     requires_grad=is_learnable)
        self.dR = torch.nn.Parameter(torch.zeros(6, dtype=torch.float32, device=self.Rt.device),
                                     requires_grad=is_learnable)
        self.register_buffer("id", torch.tensor([1, 0, 0, 0, 1, 0], device=self.Rt.device, dtype=torch.float32))
        self.register_buffer("eye4_row4", torch.tensor([0, 0, 0, 1], device=self.Rt.device, dtype=self.Rt.dtype))

    def forward(self) -> torch.Tensor:
        if not self.is_learnable:
            return self.Rt
        rot = rotation_6d_to_matrix(self.dR + self.id)
        dRt = torch.cat([torch.cat([rot, self.dt.view(3, 1)], dim=-1), self.eye4_row4.view(1, 4)])
        return torch.matmul(self.Rt, dRt)


Pose = PoseZhou  # gslam/map.py:5 imports the pose type under this name


class _PoseLink:
    """hand-over between the projection backward and the pose backward of one pose_batch() result (see ops._Projection):
    ``partials`` = (buffer, n_blocks) of per-workgroup pose-gradient partials, set by the projection backward and
    consumed by _PoseBatch.backward"""
    __slots__ = ("count", "claimed", "partials")

    def __init__(self, count: int):
        self.count, self.claimed, self.partials = count, False, None


class _PoseBatch(torch.autograd.Function):
    """viewmats [C,4,4] of C PoseZhou modules in one launch (csrc/pose.hip); inputs are (Rt_0, dR_0, dt_0, Rt_1, ...)."""

    @staticmethod
    def forward(ctx, link, learnable, *tensors):
        import ctypes as C
        from ._lib import check, lib, stream_ptr
        ctx.link = link
        ctx.set_materialize_grads(False)          # a projection that took the link returns no v_viewmats at all
        n = len(learnable)
        Rts, dRs, dts = tensors[0::3], tensors[1::3], tensors[2::3]
        dev = Rts[0].device
        out = torch.empty(n, 4, 4, dtype=torch.float32, device=dev)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        check(lib.gsx_pose_zhou_fwd(n, arr(Rts), arr(dRs), arr(dts), (C.c_int * n)(*learnable), out.data_ptr(),
                                    stream_ptr(dev)), "gsx_pose_zhou_fwd")
        ctx.learnable = learnable
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, v_view):
        import ctypes as C
        from ._lib import check, lib, stream_ptr
        tensors = ctx.saved_tensors
        learnable = ctx.learnable
        n = len(learnable)
        Rts, dRs, dts = tensors[0::3], tensors[1::3], tensors[2::3]
        link = ctx.link
        partials, link.partials = link.partials, None
        if partials is not None and partials[2] != torch._C._current_graph_task_id():
            partials = None                               # left behind by an earlier backward pass: stale
        if v_view is None and partials is None:
            return (None, None) + (None,) * len(tensors)
        dev = Rts[0].device
        v_dR = [torch.empty_like(t) for t in dRs]
        v_dt = [torch.empty_like(t) for t in dts]
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        if partials is not None:
            # the projection backward left its per-workgroup pose partials (ops._Projection.backward): summed and
            # pushed through the pose algebra in one launch; v_view holds what reached the view matrices otherwise
            buf, n_blocks = partials[0], partials[1]
            extra = None if v_view is None else v_view.contiguous()
            check(lib.gsx_pose_zhou_bwd_partials(n, arr(Rts), arr(dRs), arr(dts), (C.c_int * n)(*learnable),
                                                 buf.data_ptr(), n_blocks, None if extra is None else extra.data_ptr(),
                                                 arr(v_dR), arr(v_dt), stream_ptr(dev)), "gsx_pose_zhou_bwd_partials")
        else:
            check(lib.gsx_pose_zhou_bwd(n, arr(Rts), arr(dRs), arr(dts), (C.c_int * n)(*learnable),
                                        v_view.contiguous().data_ptr(), arr(v_dR), arr(v_dt), stream_ptr(dev)),
                  "gsx_pose_zhou_bwd")
        grads = [None, None]
        for i in range(n):
            grads += [None, v_dR[i] if learnable[i] else None, v_dt[i] if learnable[i] else None]
        return tuple(grads)


def pose_batch(poses) -> torch.Tensor:
    """== torch.stack([p() for p in poses]) for PoseZhou modules, as one HIP launch (and one for the backward)."""
    poses = list(poses)
    if (len(poses) > 16 or not all(isinstance(p, PoseZhou) for p in poses) or not poses[0].Rt.is_cuda
            or any(p.Rt.dtype != torch.float32 for p in poses)):
        return torch.stack([p() for p in poses], dim=0)
    flat = []
    for p in poses:
        flat += [p.Rt.contiguous(), p.dR, p.dt]
    link = _PoseLink(len(poses))
    out = _PoseBatch.apply(link, tuple(1 if p.is_learnable else 0 for p in poses), *flat)
    if out.requires_grad:
        out._gsx_pose_link = link
    return out


@dataclass
class Camera:
    intrinsics: torch.Tensor
    height: int
    width: int

    def to(self, device):
        self.intrinsics = self.intrinsics.to(device)
        return self

    def clone(self):
        return Camera(self.intrinsics.detach(), self.height, self.width)

    @torch.no_grad()
    def backproject(self, depth_map: torch.Tensor) -> torch.Tensor:
        """[H,W] depth -> [H*W,3] camera-space points (gslam/primitives.py:369-395)."""
        fx, fy = self.intrinsics[0, 0], self.intrinsics[1, 1]
        cx, cy = self.intrinsics[0, 2], self.intrinsics[1, 2]
        H, W = depth_map.shape
        vs, us = torch.meshgrid(torch.arange(H, device=depth_map.device), torch.arange(W, device=depth_map.device),
                                indexing='ij')
        xs = (us - cx) * (depth_map / fx)
        ys = (vs - cy) * (depth_map / fy)
        return torch.stack([xs, ys, depth_map], dim=-1).reshape(-1, 3)


@dataclass
class Frame:
    img: torch.Tensor
    timestamp: float
    camera: Camera
    pose: PoseZhou
    gt_pose: torch.Tensor
    index: int
    gt_depth: torch.Tensor = None
    img_file: str = None
    visible_gaussians: torch.Tensor = None
    est_depths: torch.Tensor = None
    exposure_params: torch.Tensor = None

    def to(self, device):
        attrs = {k: (v.to(device) if hasattr(v, 'to') else v) for k, v in vars(self).items()}
        return type(self)(**attrs)

    @torch.no_grad()
    def strip(self):
        return type(self)(None, self.timestamp, self.camera, deepcopy(self.pose), self.gt_pose, self.index, None,
                          self.img_file, None, None, self.exposure_params.detach().clone()).to(self.img.device)
