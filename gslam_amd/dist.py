"""Multi-GPU keyframe bundle adjustment: one process per GPU, ``torch.distributed`` over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).  New design - the reference is single-GPU with no collectives
(SURVEY.md §2.1, §8e).

Partitioning: the map (60 B/Gaussian params + 60 B grads + 120 B Adam state) is replicated; rank r renders the
keyframes {c : c mod G == r} of the BA window.  Per iteration there is exactly ONE data-path collective: an
all-reduce(sum) of a single contiguous fp32 bucket (``StepBucket``: map gradients, visibility counts, pose gradients,
loss terms).  xGMI is point-to-point (7 links x ~153 GB/s per GPU): one large bucket lets RCCL drive all links; many
small per-tensor all-reduces would be latency-bound.  Between iterations: all-reduce(max) of the screen radii before
size pruning (backend.py:364) and all-reduce(sum) of the densification statistic (insertion.py:298-308), both
``KeyframeShard`` helpers called by gslam_amd.backend.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as td

GRAD_PARAMS = ('means', 'quats', 'scales', 'opacities', 'colors', 'log_uncertainties')


def init_from_env(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract).
    Returns (rank, world_size); no-op (0, 1) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None and device.type == "cuda":
            kwargs["device_id"] = device
        td.init_process_group(backend=backend, **kwargs)
    return td.get_rank(), td.get_world_size()


class KeyframeShard:
    """Round-robin ownership of window keyframes by rank."""

    def __init__(self, group=None):
        self.group = group
        if td.is_available() and td.is_initialized():
            self.rank, self.world_size = td.get_rank(group), td.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1

    def select(self, window: Sequence) -> List:
        if self.world_size == 1:
            return list(window)
        return [f for i, f in enumerate(window) if i % self.world_size == self.rank]

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            td.all_reduce(t, op=td.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            td.all_reduce(t, op=td.ReduceOp.MAX, group=self.group)
        return t

    def broadcast_(self, tensors: Sequence[torch.Tensor], src: int = 0) -> None:
        if self.world_size > 1:
            for t in tensors:
                td.broadcast(t, src=src, group=self.group)


class StepBucket:
    """Everything one BA iteration sums over ranks, in ONE flat fp32 buffer:

        [ N*15 map gradients | N visible-camera counts | Cw*3 pose dt gradients | Cw*6 pose dR gradients | 2 loss values |
          overflow flag | spare ]

    (means3 + quats4 + scales3 + opac1 + colors3 + log_unc1 = 15 columns; Cw = cameras of the whole BA window).  The
    kernels of a launch plan (gslam_amd.plan.MappingStep) write their outputs straight into the views below - no
    cat / split copies - and ``reduce()`` is the single all-reduce(sum) of the iteration: it carries the map gradients,
    the per-Gaussian visibility counts (isotropic term backend.py:287, opacity decay :357; exact in fp32: counts <= C),
    the pose gradients (each rank fills the rows of the cameras it rendered, the rest are zero) and the loss terms
    (each rank's share of the window means), so every rank can apply the identical update to the map and to ALL window
    poses and take the identical early-stop decision without a second collective or a pose broadcast.  The overflow flag
    (1.0 from every rank whose render truncated a tile list this iteration, gsx_status_flag) rides in the same sum: the
    update launches are gated on it ON THE DEVICE (gsx_adam_multi_steps_gated), so either every rank applies the iteration's
    update or none does, and every rank reads the same flag next to the loss value - the decision to redo an iteration is
    collective by construction."""

    def __init__(self, shapes: Sequence[Sequence[int]], n_window_cams: int, device, group=None):
        self.group = group
        shapes = [tuple(int(x) for x in s) for s in shapes]
        self.N = n = shapes[0][0]
        self.Cw = cw = int(n_window_cams)
        numels = [int(torch.Size(s).numel()) for s in shapes]
        n_map = sum(numels)
        self.flat = torch.zeros(n_map + n + cw * 9 + 4, dtype=torch.float32, device=device)
        self.views, off = [], 0
        for s, k in zip(shapes, numels):
            self.views.append(self.flat[off:off + k].view(s))
            off += k
        self.map_part = self.flat[:n_map]
        self.counts = self.flat[off:off + n]; off += n
        self.tail = self.flat[off:]                       # pose rows + loss slots: re-zeroed every iteration when sharded
        self.g_dt = self.flat[off:off + cw * 3].view(cw, 3); off += cw * 3
        self.g_dR = self.flat[off:off + cw * 6].view(cw, 6); off += cw * 6
        self.out2 = self.flat[off:off + 2]
        self.overflow = self.flat[off + 2:off + 3]         # > 0 after the reduction: some rank's tile lists overflowed
        self.out4 = self.flat[off:off + 4]                 # (total, photometric, overflow flag, spare): one read-back
        self.vis_i32 = torch.zeros(n, dtype=torch.int32, device=device)   # window-wide counts after the reduction

    @property
    def world_size(self) -> int:
        return td.get_world_size(self.group) if (td.is_available() and td.is_initialized()) else 1

    @torch.no_grad()
    def reduce(self, local_vis: torch.Tensor | None):
        """local_vis: this rank's int32 [N] visible-camera counts, or None when it rendered no camera of the window (a
        window shorter than the world size): it then contributes zeros and still joins the collective"""
        if self.world_size == 1:
            if local_vis is not None:
                self.vis_i32.copy_(local_vis)
            return
        if local_vis is not None:
            self.counts.copy_(local_vis)
        else:
            self.map_part.zero_()
            self.counts.zero_()
        td.all_reduce(self.flat, op=td.ReduceOp.SUM, group=self.group)
        self.vis_i32.copy_(self.counts)
