"""Multi-GPU keyframe bundle adjustment: one process per GPU, ``torch.distributed`` over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).  New design - the reference is single-GPU with no collectives
(SURVEY.md §2.1, §8e).

Partitioning: the map's parameters (60 B/Gaussian) are replicated, rank r renders the keyframes {c : c mod G == r} of the BA
window, and the UPDATE is sharded over Gaussians: per iteration one small all-reduce of the "head" (visibility counts, pose
gradients, loss, overflow flag), one reduce-scatter of the flat gradient bucket (rank r receives the window-wide sum of
chunk r), Adam on that 1/G of the map (the moments live only on their owner: 120 B/Gaussian / G) and one all-gather of the
updated parameter chunks (``StepBucket``).  Same bytes on the wire as an all-reduce of the bucket, optimiser time / G.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): two large collectives let RCCL drive all links; per-tensor
collectives would be latency-bound.  Between iterations: all-reduce(max) of the screen radii before
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


def bucket_layout(shapes: Sequence[Sequence[int]], world: int):
    """tensor-major layout of the six per-Gaussian arrays in ONE flat fp32 buffer: every array starts on a 16-byte boundary
    and the whole is padded to ``world`` chunks of equal length L (a multiple of 4 floats).  The parameters, their gradients
    and the two Adam moments all use this layout, so chunk r of one is chunk r of the others.
    -> (offsets, numels, S_pad, L)"""
    offs, numels, off = [], [], 0
    for s in shapes:
        k = int(torch.Size(tuple(int(x) for x in s)).numel())
        offs.append(off)
        numels.append(k)
        off += (k + 3) // 4 * 4
    unit = 4 * max(int(world), 1)
    s_pad = (off + unit - 1) // unit * unit
    return offs, numels, s_pad, s_pad // max(int(world), 1)


def ranged_layout(shapes: Sequence[Sequence[int]], world: int, ranges: int):
    """Layout of the RANGED exchange: the map's rows are cut into K ranges of Nr rows; Nr is a multiple of 256 (a range is a
    whole number of workgroups of the projection backward, gsx_project_bwd_range) and of 4 * world (every rank's part of every
    array of a range starts on a 16-byte boundary); every array is padded to N_pad = K * Nr rows inside the flat buffer (the pad
    rows are zero everywhere and stay zero: zero gradient, zero moments).  K <= ``ranges`` (a small map has fewer).
    -> (offsets, numels, S_pad, L, Nr, K, dims)"""
    import math
    world = max(int(world), 1)
    n = int(shapes[0][0])
    numels = [int(torch.Size(tuple(int(x) for x in s)).numel()) for s in shapes]
    dims = [k // max(n, 1) if n else 1 for k in numels]
    if any(d * n != k for d, k in zip(dims, numels)):
        raise ValueError("ranged exchange: every array needs the map's row count as its first dimension")
    unit = 256 * (4 * world) // math.gcd(256, 4 * world)
    per = (max(n, 1) + max(int(ranges), 1) - 1) // max(int(ranges), 1)
    nr = (per + unit - 1) // unit * unit
    k_eff = max(1, (n + nr - 1) // nr)
    n_pad = k_eff * nr
    offs, off = [], 0
    for d in dims:
        offs.append(off)
        off += n_pad * d
    return offs, numels, off, off // world, nr, k_eff, dims


class StepBucket:
    """What one BA iteration exchanges between ranks (new design: the reference is single-GPU).

    ``flat`` - the map gradients of this rank's cameras, tensor-major (``bucket_layout``): means3 | quats4 | scales3 |
    opac1 | colors3 | log_unc1 per Gaussian, 15 N floats.  The kernels of a launch plan (gslam_amd.plan.MappingStep) write
    straight into the views; between ranks it goes through ONE reduce-scatter: rank r receives the window-wide sum of chunk r
    (``gchunk``), runs Adam on that 1/G of the map (its slices of the parameters and of the moments) and the updated
    parameter chunks come back to everybody through ONE all-gather - the same bytes on the wire as an all-reduce of the
    bucket, the optimiser's 28 B per parameter divided by G, and moments that are only ever valid on their owner.

    ``head`` - what every rank needs whole, in one small all-reduce BEFORE the reduce-scatter: the per-Gaussian
    visible-camera counts (isotropic term backend.py:287, opacity decay :357; exact in fp32, counts <= C), the pose
    gradients of all window cameras (each rank fills its cameras' rows; every rank then applies the identical pose update),
    the loss values (each rank's share of the window means) and the overflow flag (1.0 from every rank whose render
    truncated a tile list: the update launches are gated on the sum ON THE DEVICE, so every rank applies the iteration or
    none does and they all read the same flag with the loss - the decision to redo an iteration is collective by
    construction).   [ N counts | Cw*3 dt | Cw*6 dR | total, photometric | overflow, spare ]

    One rank: no collective; ``flat`` and the head's tail are simply where the kernels leave their results.

    ``ranges`` = K > 0: the RANGED exchange (``ranged_layout``).  The map's rows are cut into K ranges; the exchange of range k
    is one reduce-scatter (and one all-gather) of ITS rows of all six arrays, so it can be issued as soon as the projection
    backward has finished those rows and run beside the rest of the backward (``reduce_counts / reduce_range / reduce_tail /
    gather_range``: gslam_amd.plan.MappingStep drives them on a second stream).  Rank r owns part r of every array of every
    range - 6 K pieces instead of one chunk; Adam is elementwise, so who owns an element changes nothing in its value.  The
    arrays stay tensor-major (the kernels address rows); a range's parts are gathered into one staging block per rank in front
    of the collective (gsx_range_copy: one launch) and scattered back behind the all-gather.  ``reduce`` / ``gather`` keep their
    meaning in this layout (all ranges, one after the other): everything written against the one-shot bucket works on both."""

    def __init__(self, shapes: Sequence[Sequence[int]], n_window_cams: int, device, group=None, world: int | None = None,
                 rank: int | None = None, ranges: int = 0):
        """world / rank: default = those of ``group`` (the default process group); world = 1 makes a local bucket whatever
        process group exists (the single-rank reference of the multi-rank tests)"""
        self.group = group
        shapes = [tuple(int(x) for x in s) for s in shapes]
        self.N = n = shapes[0][0]
        self.Cw = cw = int(n_window_cams)
        self.world = self.world_size if world is None else int(world)
        self.rank = (td.get_rank(group) if self.world > 1 else 0) if rank is None else int(rank)
        self.ranges = 0
        if int(ranges) > 0 and self.world > 1:
            self.offsets, self.numels, self.S_pad, self.L, self.Nr, self.ranges, self.dims = ranged_layout(
                shapes, self.world, int(ranges))
            self.part_len = [self.Nr * d // self.world for d in self.dims]      # floats of one rank's part of array t in a range
            self.Lk = sum(self.part_len)                                        # one rank's block of a range
            self.stage_rs = torch.zeros(self.world * self.Lk, dtype=torch.float32, device=device)
            self.stage_in = torch.zeros(self.Lk, dtype=torch.float32, device=device)
            self.stage_out = torch.zeros(self.world * self.Lk, dtype=torch.float32, device=device)
            self._copy_args = {}
        else:
            self.offsets, self.numels, self.S_pad, self.L = bucket_layout(shapes, self.world)
        self.flat = torch.zeros(self.S_pad, dtype=torch.float32, device=device)
        self.views = [self.flat[o:o + k].view(s) for s, o, k in zip(shapes, self.offsets, self.numels)]
        self.map_part = self.flat
        self.head = torch.zeros(n + cw * 9 + 4, dtype=torch.float32, device=device)
        self.counts = self.head[:n]
        off = n
        self.tail = self.head[off:]                       # pose rows + loss slots + flag: re-zeroed every iteration when sharded
        self.g_dt = self.head[off:off + cw * 3].view(cw, 3); off += cw * 3
        self.g_dR = self.head[off:off + cw * 6].view(cw, 6); off += cw * 6
        self.out2 = self.head[off:off + 2]
        self.overflow = self.head[off + 2:off + 3]        # > 0 after the reduction: some rank's tile lists overflowed
        self.out4 = self.head[off:off + 4]                # (total, photometric, overflow flag, spare): one read-back
        # window-wide counts after the reduction (ranged layout: with the pad rows, whose count stays 0)
        self.vis_all = torch.zeros(self.ranges * self.Nr if self.ranges else n, dtype=torch.int32, device=device)
        self.vis_i32 = self.vis_all[:n]
        # this rank's chunk of the summed gradients (what its slice of Adam reads)
        self.gchunk = torch.zeros(self.L, dtype=torch.float32, device=device) if self.world > 1 else None

    @property
    def world_size(self) -> int:
        return td.get_world_size(self.group) if (td.is_available() and td.is_initialized()) else 1

    def chunk_range(self, rank: int | None = None):
        if self.ranges:
            raise RuntimeError("the ranged layout has no single chunk per rank: use pieces()")
        r = self.rank if rank is None else int(rank)
        return r * self.L, (r + 1) * self.L

    def pieces_of_range(self, k: int, rank: int | None = None):
        """the six pieces rank owns of range k (ranged layout), as ``pieces`` describes them"""
        r = self.rank if rank is None else int(rank)
        out, poff = [], 0
        for t, (d, pl) in enumerate(zip(self.dims, self.part_len)):
            out.append((t, k * self.Nr * d + r * pl, pl, k * self.Lk + poff))
            poff += pl
        return out

    def pieces(self, rank: int | None = None):
        """[(tensor index k, start inside tensor k, length, start inside the chunk)] - the slices of the six arrays that
        rank's chunk covers (boundaries fall on multiples of 4 floats: 16-byte accesses everywhere).  Ranged layout: the
        6 K pieces of the rank, range by range; starts count in the PADDED arrays (N_pad rows), "the chunk" is ``gchunk``"""
        if self.ranges:
            return [pc for k in range(self.ranges) for pc in self.pieces_of_range(k, rank)]
        lo, hi = self.chunk_range(rank)
        out = []
        for k, (o, n) in enumerate(zip(self.offsets, self.numels)):
            a, b = max(lo, o), min(hi, o + n)
            if b > a:
                out.append((k, a - o, b - a, a - lo))
        return out

    @torch.no_grad()
    def reduce(self, local_vis: torch.Tensor | None, between=None):
        """The gradient exchange of one iteration: all-reduce of the head, then - with the window-wide visibility known -
        ``between()`` (the plan adds what depends on it to the LOCAL gradients: the isotropic term, on rank 0 alone so that
        it enters the sum once), then the reduce-scatter of the gradient bucket into ``gchunk``.
        local_vis: this rank's int32 [N] visible-camera counts, or None when it rendered no camera of the window (a window
        shorter than the world size): it then contributes zeros and still joins the collectives"""
        if self.world == 1:
            if local_vis is not None:
                self.vis_i32.copy_(local_vis)
            return
        if local_vis is not None:
            self.counts.copy_(local_vis)
        else:
            self.flat.zero_()
            self.counts.zero_()
        td.all_reduce(self.head, op=td.ReduceOp.SUM, group=self.group)
        self.vis_i32.copy_(self.counts)
        if between is not None:
            between()
        if self.ranges:
            for k in range(self.ranges):
                self.reduce_range(k)
            return
        td.reduce_scatter_tensor(self.gchunk, self.flat, op=td.ReduceOp.SUM, group=self.group)

    # ---- the ranged exchange, piece by piece (every call works on torch's CURRENT stream) --------------------------------------
    def _range_copy(self, whole: torch.Tensor, k: int, staging: torch.Tensor, own_only: bool, to_flat: bool):
        """between range k's rows of the six arrays inside ``whole`` (a flat buffer of the bucket's layout) and a staging buffer:
        all ranks' parts <-> ``world`` blocks, or (own_only) this rank's parts <-> one block"""
        parts = 1 if own_only else self.world
        starts = [o + k * self.Nr * d + (self.rank * pl if own_only else 0)
                  for o, d, pl in zip(self.offsets, self.dims, self.part_len)]
        if whole.is_cuda:
            import ctypes as C
            from ._lib import check, lib
            key = (whole.data_ptr(), k, own_only)
            args = self._copy_args.get(key)
            if args is None:
                n = len(starts)
                args = ((C.c_void_p * n)(*[whole.data_ptr() + 4 * a for a in starts]), (C.c_int64 * n)(*self.part_len))
                self._copy_args[key] = args
            check(lib.gsx_range_copy(len(starts), args[0], args[1], parts, staging.data_ptr(), 1 if to_flat else 0,
                                     torch.cuda.current_stream(whole.device).cuda_stream), "gsx_range_copy")
            return
        # host tensors (the gloo tests of the layout): the same copy with views
        blocks, poff = staging.view(parts, self.Lk), 0
        for a, pl in zip(starts, self.part_len):
            src = whole[a:a + parts * pl].view(parts, pl)
            if to_flat:
                src.copy_(blocks[:, poff:poff + pl])
            else:
                blocks[:, poff:poff + pl].copy_(src)
            poff += pl

    @torch.no_grad()
    def reduce_counts(self, local_vis: torch.Tensor | None):
        """all-reduce of the visible-camera counts alone (they are known after the forward projection)"""
        if local_vis is not None:
            self.counts.copy_(local_vis)
        else:
            self.counts.zero_()
        td.all_reduce(self.counts, op=td.ReduceOp.SUM, group=self.group)
        self.vis_i32.copy_(self.counts)

    @torch.no_grad()
    def reduce_tail(self):
        """all-reduce of the head's tail: pose rows, loss slots, overflow flag"""
        td.all_reduce(self.tail, op=td.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def reduce_range(self, k: int):
        """range k of the local gradients -> this rank's block of their window-wide sum (``gchunk[k * Lk : (k + 1) * Lk]``)"""
        self._range_copy(self.flat, k, self.stage_rs, own_only=False, to_flat=False)
        td.reduce_scatter_tensor(self.gchunk[k * self.Lk:(k + 1) * self.Lk], self.stage_rs, op=td.ReduceOp.SUM,
                                 group=self.group)

    @torch.no_grad()
    def gather_range(self, whole: torch.Tensor, k: int):
        """range k of a flat buffer whose pieces are valid on their owners -> whole on every rank"""
        self._range_copy(whole, k, self.stage_in, own_only=True, to_flat=False)
        td.all_gather_into_tensor(self.stage_out, self.stage_in, group=self.group)
        self._range_copy(whole, k, self.stage_out, own_only=False, to_flat=True)

    @torch.no_grad()
    def gather(self, whole: torch.Tensor, staging: torch.Tensor):
        """all-gather of a chunked flat buffer (updated parameters after the sharded Adam; the moments before the map is
        re-packed): ``whole`` [S_pad] holds this rank's valid chunk, every rank ends with all of it.  ``staging`` [L]."""
        if self.world == 1:
            return
        if self.ranges:
            for k in range(self.ranges):
                self.gather_range(whole, k)
            return
        lo, hi = self.chunk_range()
        staging.copy_(whole[lo:hi])
        td.all_gather_into_tensor(whole, staging, group=self.group)


def flatten_map_state(splats, opt, bucket: StepBucket):
    """Re-homes the six per-Gaussian parameter tensors of ``splats`` and their Adam moments (``opt``: the FusedAdam that owns
    them) into three flat buffers of the bucket's layout - the tensors keep their shapes and become views - so that chunk r of
    the parameters, of the gradients and of both moments is the same slice of the map.  Idempotent while the map's tensors
    stay where this put them (every launch plan over the same map shares the buffers); a re-packed map (pruning, insertion)
    has new tensors and is flattened again by its next plan.  -> dict(pflat, mflat, vflat, stage, sharded)"""
    params = [getattr(splats, n) for n in GRAD_PARAMS]
    st = getattr(splats, "_gsx_flat", None)
    if st is not None and st["S_pad"] == bucket.S_pad and st["offsets"] == bucket.offsets and all(
            p.data_ptr() == st["pflat"].data_ptr() + 4 * o and p.numel() == k
            for p, o, k in zip(params, bucket.offsets, bucket.numels)):
        return st
    dev = params[0].device
    pflat = torch.zeros(bucket.S_pad, dtype=torch.float32, device=dev)
    mflat, vflat = torch.zeros_like(pflat), torch.zeros_like(pflat)
    with torch.no_grad():
        for p, o, k in zip(params, bucket.offsets, bucket.numels):
            if not (p.is_cuda and p.dtype == torch.float32):
                raise RuntimeError("sharded update needs float32 map tensors on the GPU")
            view = pflat[o:o + k].view(p.shape)
            view.copy_(p.data)
            p.data = view
            state = opt.state[p]
            for name, flat in (("exp_avg", mflat), ("exp_avg_sq", vflat)):
                mv = flat[o:o + k].view(p.shape)
                if name in state:
                    mv.copy_(state[name])
                state[name] = mv
    st = dict(pflat=pflat, mflat=mflat, vflat=vflat, stage=torch.zeros(bucket.L, dtype=torch.float32, device=dev),
              S_pad=bucket.S_pad, offsets=list(bucket.offsets), sharded=False)
    splats._gsx_flat = st
    return st
