"""Launch plans: the closures of the tracking and mapping optimisers as fixed chains of libgsx launches over persistent
device buffers, recorded once into a HIP graph (csrc/runtime.hip) and replayed.

The reference evaluates every closure from Python through autograd - ~60 kernel launches, a dozen allocations and a
``loss.item()`` per evaluation (gslam/frontend.py:621-649, gslam/backend.py:260-359,465-504).  The autograd-shaped
operators of this package (``gslam_amd.rasterization`` etc.) keep that interface for drop-in use; the optimisation loops
themselves run on the plans below, which issue the SAME C-ABI calls in the same order with every buffer allocated up
front:

* no autograd graph, no AccumulateGrad nodes, no allocator traffic inside the closure - nothing that can tie a captured
  graph to the stream a tensor was first used on, and nothing that is allocated or freed while a stream captures;
* capture / instantiate / launch are plain HIP calls on a stream the plan owns (``HipGraph``);
* sizes stay on the device: the tile-list capacity is a property of the plan, overflow is a sticky device status word
  that ``check_capacity()`` reads once per frame / refinement / BA round, and a plan that overflowed grows its buffers and
  is re-captured by its owner.

``RenderPlan`` is one differentiable render of a fixed shape (N, C, W, H, CH); ``TrackClosure`` (tracking,
frontend.py:604-662), ``WindowClosure`` (window pose refinement, backend.py:447-506) and ``MappingStep`` (one BA
iteration, backend.py:260-359) put the loss, the pose algebra and the optimiser step around it."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import torch

from ._lib import check, lib
from .ops import PROJ_BETAS, PROJ_LOG_SCALES, PROJ_RENDER_DEPTH

TILE = 16
_VIEW_PARTIALS = 8      # GSX_PROJ_VIEW_PARTIALS
_RESET_V_REC = 64       # GSX_PROJ_RESET_V_REC
_SKIP_CULLED = 16       # GSX_PROJ_SKIP_CULLED
_COMPACT = 32           # GSX_PROJ_COMPACT
_CANDIDATES = 128       # GSX_PROJ_CANDIDATES
_DEFER_SORT = 256       # GSX_PROJ_DEFER_SORT
_MAP_RECORDS = 512      # GSX_PROJ_MAP_RECORDS
_ROW_KEYS = 1024        # GSX_PROJ_ROW_KEYS
_TILE_EXACT = 2048      # GSX_PROJ_TILE_EXACT


def _arr(ptrs: Sequence[Optional[int]]):
    return (C.c_void_p * len(ptrs))(*ptrs)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def current_stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class HipGraph:
    """An instantiated hipGraph of one launch chain (csrc/runtime.hip)."""

    def __init__(self):
        self._exec = None
        self.nodes = 0

    @property
    def captured(self) -> bool:
        return self._exec is not None

    def capture(self, stream: torch.cuda.Stream, enqueue, mode: int = 1):
        """records ``enqueue(stream_ptr)`` - kernel launches only - from ``stream`` (never the legacy default stream)"""
        st = stream.cuda_stream
        if st == 0:
            raise RuntimeError("the default stream cannot capture; pass a plan-owned stream")
        self.destroy()
        check(lib.gsx_graph_begin(st, mode), "gsx_graph_begin")
        try:
            enqueue(st)
        except BaseException:
            lib.gsx_graph_abort(st)
            raise
        ex, n = C.c_void_p(), C.c_int64(0)
        check(lib.gsx_graph_end(st, C.byref(ex), C.byref(n)), "gsx_graph_end")
        self._exec, self.nodes = ex, int(n.value)

    def launch(self, stream_ptr: Optional[int] = None, count: int = 1):
        if self._exec is None:
            raise RuntimeError("graph not captured")
        st = current_stream_ptr() if stream_ptr is None else stream_ptr
        if count == 1:
            check(lib.gsx_graph_launch(self._exec, st), "gsx_graph_launch")
        else:
            check(lib.gsx_graph_launch_n(self._exec, int(count), st), "gsx_graph_launch_n")

    def replay(self):
        """one launch on torch's current stream (the name torch.cuda.CUDAGraph uses)"""
        self.launch()

    def destroy(self):
        if self._exec is not None:
            ex, self._exec = self._exec, None
            lib.gsx_graph_destroy(ex)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _launch_chain(plan, count: int, stream_ptr: Optional[int] = None):
    """``count`` evaluations of ``plan``'s closure as ONE graph launch: consecutive launches of a graph sit ~9 us apart on the
    device (traced), nodes inside one ~1.5 us - a frame of 36 closures loses a third of a millisecond between launches of the
    one-closure graph.  The chain of ``count`` closures is captured on first use and kept until the closure is re-captured."""
    if count <= 1:
        return plan.graph.launch(stream_ptr, count)
    g = plan._chains.get(count)
    if g is None:
        from ._sync import capture_lock
        g = HipGraph()

        def enqueue(st):
            for _ in range(count):
                plan.enqueue(st)
        with capture_lock:
            plan.stream.wait_stream(torch.cuda.current_stream(plan.dev))
            g.capture(plan.stream, enqueue)
            torch.cuda.current_stream(plan.dev).wait_stream(plan.stream)
        plan._chains[count] = g
    g.launch(stream_ptr)


def _drop_chains(plan):
    for g in plan._chains.values():
        g.destroy()
    plan._chains.clear()


def _map_tensors(splats):
    ts = [splats.means, splats.quats, splats.scales, splats.opacities, splats.colors, splats.log_uncertainties]
    out = []
    for t in ts:
        d = t.detach()
        if not (d.is_cuda and d.dtype == torch.float32 and d.is_contiguous()):
            raise RuntimeError("plans need contiguous float32 map tensors on the GPU (no CPU fallback)")
        out.append(d)
    return out


_PLACEMENT: Dict[tuple, tuple] = {}
_PLACEMENT_FAILS: Dict[tuple, int] = {}
PLACEMENT_RETRIES = 3       # a failed probe is repeated by later plans this many times before the verdict sticks


def raster_lds_bytes() -> int:
    """static LDS of a workgroup of the fused tracking rasteriser, asked of the loaded code object (it depends on GSX_TC_ROWS,
    the staging strides and the tile-sort pool, which diagnostic builds override): what the placement probe's workgroups allocate,
    so that they are placed like the launches whose order the probe vouches for"""
    n = int(lib.gsx_raster_track_fused_lds_bytes())
    if n <= 0:
        raise RuntimeError("gsx_raster_track_fused_lds_bytes failed: " + lib.gsx_last_error().decode(errors="replace"))
    return n


def placement_ok(dev, n_wgs: int, n_cus: int, lds_bytes: Optional[int] = None):
    """Does workgroup i of an ``n_wgs``-workgroup launch (256 threads, the rasteriser's LDS footprint, all resident at once)
    share its compute unit with workgroups i + G, i + 2 G, ... on this device?  The CU-balanced launch order
    (csrc/tile_balance.h) deals the tiles into G groups on that assumption - an undocumented property of the dispatcher,
    traced on MI355X / ROCm 7.2.  Probed on a drained chip (gsx_probe_wg_placement, ~50 us).  A POSITIVE result
    is cached per (device, shape).  A negative one is not trusted at once: the probe needs the chip to itself and
    ``torch.cuda.synchronize`` drains it only for an instant - another host thread (the backend's BA stream) can enqueue work
    before the probe's workgroups are placed - so the probe is repeated twice on the spot and, if it still fails, by the next
    ``PLACEMENT_RETRIES`` plans that ask; only then does the identity / heaviest-first order stick (logged).  -> (ok, note)"""
    if lds_bytes is None:
        lds_bytes = raster_lds_bytes()
    key = (str(dev), int(n_wgs), int(n_cus), int(lds_bytes))
    hit = _PLACEMENT.get(key)
    if hit is not None:
        return hit
    if torch.cuda.is_current_stream_capturing():
        return False, "placement probe skipped (stream is capturing)"
    idx = torch.arange(n_wgs) % n_cus
    res = None
    for _attempt in range(3):
        torch.cuda.synchronize(dev)
        keys = torch.full((n_wgs,), -1, dtype=torch.int32, device=dev)
        check(lib.gsx_probe_wg_placement(int(n_wgs), int(lds_bytes), 30, _p(keys), current_stream_ptr(dev)),
              "gsx_probe_wg_placement")
        k = keys.cpu()
        first = k[:n_cus]
        distinct = int(torch.unique(first).numel())
        if distinct == n_cus and bool((k == first[idx]).all()):
            res = (True, f"cu-balanced (placement probe ok: {n_wgs} workgroups on {n_cus} compute units, i and i + {n_cus} "
                         "share one)")
            break
        bad = int((k != first[idx]).sum())
        res = (False, f"identity order (placement probe FAILED: {distinct} distinct compute units among the first {n_cus} "
                      f"workgroups, {bad} of {n_wgs} workgroups off the i mod {n_cus} pattern)")
    if res[0]:
        _PLACEMENT[key] = res
        return res
    fails = _PLACEMENT_FAILS.get(key, 0) + 1
    _PLACEMENT_FAILS[key] = fails
    if fails > PLACEMENT_RETRIES:
        _PLACEMENT[key] = res
        import warnings
        warnings.warn("gslam_amd: " + res[1])
    return res


class RenderPlan:
    """One render of gslam ``rasterization()`` (gslam/rasterization.py:44-360 with the live argument set of
    gslam/map.py:88-103) and its backward, as launches over buffers owned by the plan.

    grads: 'none' (forward only), 'pose' (frozen map: geometry-only rasteriser backward when no depth channel is rendered,
    pose-only projection backward that leaves its per-workgroup pose partials for a fused consumer), 'full' (gradients
    of all six map arrays written to ``self.v_*`` / the tensors given in ``grad_out``, plus the pose partials)."""

    GROW = 1.5
    ORDER_MAX_PER_TILE = 1000       # heaviest-first launch order (by list length) below this capacity per tile
    # MappingStep / WindowClosure on the generic chain: the projection packs every instance's TIGHT tile rectangle into one word
    # (gsx_project_fwd_rects | GSX_PROJ_TILE_EXACT) and the binning reads that instead of means2d + radius
    # (gsx_isect_bin_sort_rects).  Same render, same gradients (tests/test_gpu_plans.py); measured (tools/dbg/ab_tight.sh, same box):
    # BA iteration 500 k x 8 1106.4 -> 1079.6 us (84 % of the keys), 2 M x 8 2213.3 -> 2099.9 us; the refiner's closure 866.0 ->
    # 821.7 us and 1902 -> 1713 us.  (First form - the binning itself tightening from the 48-byte records: 32 B more per instance,
    # twice - was slower than doing nothing: 1124.7 against 1115.3 us, 2303.9 against 2212.1.)  RECT_LISTS: the packed REFERENCE
    # rectangles (the lists gsplat builds; 4 bytes read for 12): 1100.1 / 2177.0 us - what TIGHT_LISTS = False leaves on.
    TIGHT_LISTS = True
    # with packed rectangles the projection can write radii = 0 / rects = 0 for a culled row and nothing else (lean_rows; as_output()
    # restores the reference's zeros with one projection launch on demand).  Built, exact, OFF: measured SLOWER, same box - BA
    # iteration 2 M x 8 2177.5 against 2123.1 us, 500 k x 8 1048.7 against 1038.6: three quarters of the rows not written turn the
    # projection's full-line streaming stores into scattered 48-byte ones (partial lines at the memory side cost more than zeros).
    LEAN_ROWS = False
    RECT_LISTS = True               # the binning reads packed rectangles the projection wrote (4 bytes for 12; the reference's lists)

    def __init__(self, splats, n_cams: int, width: int, height: int, *, render_depth: bool, grads: str = 'pose',
                 Ks: Optional[torch.Tensor] = None, capacity: Optional[int] = None, need_n_touched: bool = False,
                 visibility_min_T: float = 0.5, grad_out: Optional[Dict[str, torch.Tensor]] = None,
                 near_plane: float = 0.01, far_plane: float = 1e10, eps2d: float = 0.3, front: Optional[bool] = None):
        """front: use the fused projection + tile-list front (gsx_front_fwd, four launches) instead of gsx_project_fwd +
        gsx_isect_bin_sort (seven); None = where it is the faster one (measured)"""
        assert grads in ('none', 'pose', 'full')
        self.splats = splats
        self.map = _map_tensors(splats)
        dev = self.map[0].device
        self.dev = dev
        self.N = N = int(self.map[0].shape[0])
        self.C = Cn = int(n_cams)
        self.W, self.H = int(width), int(height)
        self.tile_w, self.tile_h = math.ceil(self.W / TILE), math.ceil(self.H / TILE)
        self.T = Cn * self.tile_w * self.tile_h
        self.grads = grads
        self.near, self.far, self.eps2d, self.vis_min_T = float(near_plane), float(far_plane), float(eps2d), float(visibility_min_T)
        self.flags = PROJ_LOG_SCALES | PROJ_BETAS | (PROJ_RENDER_DEPTH if render_depth else 0)
        self.CH = 3 + (1 if render_depth else 0) + 1
        self.depth_index = 3 if render_depth else -1
        self.betas_index = self.CH - 1
        # frozen map and no depth channel: only the xy / conic columns of the gradient records are consumed
        self.geom_only = grads == 'pose' and not render_depth
        f32, i32 = torch.float32, torch.int32
        e = lambda *s, dtype=f32: torch.empty(*s, dtype=dtype, device=dev)
        self.viewmats = torch.eye(4, device=dev).repeat(Cn, 1, 1).contiguous()
        self.Ks = e(Cn, 3, 3)
        if Ks is not None:
            self.Ks.copy_(Ks.reshape(-1, 3, 3).expand(Cn, 3, 3))
        self.radii, self.tiles = e(Cn, N, dtype=i32), e(Cn, N, dtype=i32)
        self.means2d, self.depths = e(Cn, N, 2), e(Cn, N)        # (the conics are columns 2..4 of the record: no array of their own)
        self.vis_count = e(N, dtype=i32)
        self.rec = e(Cn, N, 12)
        # gradient records [C,N,12]: all zeros between backward passes - the projection backward zeroes every row it has
        # read (GSX_PROJ_RESET_V_REC), so no forward clears them (192 of the 416 MB the projection of a 500 k x 8 window
        # wrote: 86 -> 51 us, the backward pays 18 us of it back)
        self.v_rec = torch.zeros(Cn, N, 12, dtype=f32, device=dev) if grads != 'none' else None
        self._v_rec_dirty = False      # a backward(keep=True) left its gradients in v_rec
        self.offsets = torch.zeros(self.T + 1, dtype=i32, device=dev)
        self.M_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.status = torch.zeros(1, dtype=i32, device=dev)
        self.render, self.alphas = e(Cn, self.H, self.W, self.CH), e(Cn, self.H, self.W, 1)
        self.last_ids = e(Cn, self.H, self.W, dtype=i32)
        self.n_touched = torch.zeros(Cn, N, dtype=i32, device=dev) if need_n_touched else None
        bg = torch.zeros(Cn, self.CH, device=dev)                 # [0,0,0] (+0 depth) + e^1 for beta (rasterization.py:236-255)
        bg[:, self.betas_index] = math.e
        self.backgrounds = bg
        self.v_render = e(Cn, self.H, self.W, self.CH) if grads != 'none' else None
        self.pose_ws = None
        if grads != 'none':
            self.pose_ws = torch.empty(int(lib.gsx_project_bwd_workspace_bytes(N, Cn)), dtype=torch.uint8, device=dev)
            self.pose_blocks = int(lib.gsx_project_bwd_blocks(N))
        self.v_map: Optional[List[torch.Tensor]] = None
        if grads == 'full':
            names = ('means', 'quats', 'scales', 'opacities', 'colors', 'log_uncertainties')
            self.v_map = []
            for name, t in zip(names, self.map):
                g = None if grad_out is None else grad_out.get(name)
                if g is None:
                    g = torch.empty_like(t)
                if not (g.is_contiguous() and g.shape == t.shape and g.dtype == f32 and g.device == dev):
                    raise RuntimeError(f"grad_out[{name}] must be a contiguous float32 tensor shaped like the parameter")
                self.v_map.append(g)
        fits = self.T * 4 + 64 + 16384 <= 65536 and Cn <= 255 and 1 <= N <= 5_000_000
        self.front = (fits and Cn * N < (1 << 20)) if front is None else (bool(front) and fits)
        # a pose-only closure never reads the rows of culled instances nor the separate means2d / depths / conics arrays
        self.lean = grads == 'pose'
        self.compact = False
        if self.front and grads == 'pose':
            # pose gradient over the visible instances the front leaves behind: one partial row per (front row, camera)
            self.pose_blocks = int(lib.gsx_front_rows(N, Cn, self.tile_w, self.tile_h))
            # records and gradient records live per visible INSTANCE (slot-indexed, written densely by the projection's
            # workgroups) instead of per flatten id: a third of a 500 k map is visible, and 48-byte rows scattered over a
            # [N,12] array leave as partial-line writes; self.flat then carries slots (same order: GSX_PROJ_COMPACT)
            self.compact = True
            lay = (C.c_int64 * 4)()
            check(lib.gsx_front_layout(N, Cn, self.tile_w, self.tile_h, 4096, lay), "gsx_front_layout")
            self.front_rows, self.front_seg = int(lay[0]), int(lay[1])
            n_slots = Cn * self.front_rows * self.front_seg
            self.rec = torch.empty(n_slots, 12, dtype=f32, device=dev)
            self.v_rec = torch.zeros(n_slots, 12, dtype=f32, device=dev)
        self.capacity = 0
        self.flat = self.tile_order = self.isect_ws = None
        # per-frame candidate set (gsx_front_candidates): enable_candidates() + build_candidates() per frame
        self.candidates = False
        self.map_records = False
        self.cand_margins = (0.0, 0.0)
        # CU-balanced launch order (gsx_tile_balance): a render whose T workgroups are all resident at once (more than one
        # and at most five per CU) runs as long as its most loaded CU; enable_balance() makes the rasteriser launches follow
        # an order computed from the work the tiles took in an earlier closure
        self.tile_work = self.balanced_order = None
        # tile sort inside the fused tracking rasteriser (gsx_raster_track_fused_sorting): enable_defer_sort()
        self.defer_sort = False
        self.lean_rows = False        # generic chain with packed rectangles: culled rows get radii = 0, rects = 0 and nothing else
        self.rects = None             # uint32 [C,N]: the projection's packed tile rectangles (tight_lists / rect_lists)
        self.rect_lists = False       # generic chain: the binning reads the projection's packed rectangles (reference squares; same lists)
        self.tight_lists = False      # generic chain: tight rectangles packed by the projection (gsx_project_fwd_rects -> gsx_isect_bin_sort_rects)
        self.tile_exact = False       # fused front: the instance's tiles are those of its alpha >= 1/255 box inside the 3-sigma square
        self.row_keys = False         # the front ends with the projection; the rasteriser's tiles collect their keys (enable_row_keys)
        self._rows_last = False       # the last front ran with row keys (M is the sum of self.key_counters then)
        self.key_counters = None
        self.tile_span = None
        self.near_place = False
        self.cut_margin = self.CUT_MARGIN
        self.tile_cut = self.tile_near = self.sort_stats = self.tile_placed = None
        self.balance_note = "identity / heaviest-first order (shape does not qualify for the balanced order)"
        self.n_cus = int(torch.cuda.get_device_properties(dev).multi_processor_count) if dev.type == 'cuda' else 0
        self.last_M = 0
        self.stale = False          # set when the buffers were re-allocated: graphs over the old ones must be re-captured
        if capacity is not None:
            self._alloc_lists(int(capacity))

    # ---- tile-list capacity ------------------------------------------------------------------------------------------
    def _alloc_lists(self, capacity: int):
        self.capacity = cap = max(int(capacity), 4096)
        dev = self.dev
        self.flat = torch.empty(cap, dtype=torch.int32, device=dev)
        # heaviest-first launch order pays off while the tile lists are short (see rasterization.rasterization)
        self.tile_order = torch.empty(self.T, dtype=torch.int32, device=dev) if cap < self.ORDER_MAX_PER_TILE * self.T else None
        if self.front and self.candidates:
            nbytes = int(lib.gsx_front_workspace_bytes_cand(self.N, self.C, self.tile_w, self.tile_h, cap))
        elif self.front:
            nbytes = int(lib.gsx_front_workspace_bytes(self.N, self.C, self.tile_w, self.tile_h, cap))
        else:
            nbytes = int(lib.gsx_isect_bin_workspace_bytes_n(self.C, self.N, self.tile_w, self.tile_h, cap))
        self.isect_ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        if self.front and self.candidates:
            # an all-zero header never validates (reference rotation 0): closures take the full path until
            # build_candidates() has run over THIS workspace
            self._cand_header().zero_()
        self.stale = True

    # ---- per-frame candidate set ---------------------------------------------------------------------------------------
    def _cand_flags(self) -> int:
        return (_CANDIDATES | (_MAP_RECORDS if self.map_records else 0)) if self.candidates else 0

    def enable_map_records(self) -> bool:
        """Pose-only plans on the fused front: ``build_candidates`` leaves the pose-independent record of EVERY Gaussian (mean,
        world covariance, activated opacity / colour / beta: 64 bytes) and a packed cull row each in the workspace; the
        closures keep their own cull and read one record per survivor (GSX_PROJ_MAP_RECORDS).  Valid for any pose; to be
        rebuilt when the map's arrays change (the tracker does it once per frame: one pass over the map against 36 closures).
        Results are identical to the plain path bit for bit.  -> whether this plan's shape qualifies."""
        if not self.enable_candidates(0.0, 0.0):
            return False
        self.map_records = True
        return True

    def enable_candidates(self, rot_max: float = 0.02, trans_max: float = 0.02) -> bool:
        """Pose-only plans on the fused front: closures read the candidate set that ``build_candidates`` leaves in the
        workspace (the Gaussians that can be visible from any pose within ``rot_max`` (|R R0^T - I|_F) and ``trans_max`` of the
        reference poses, with their pose-independent projection inputs) instead of culling the whole map again; a closure
        whose poses left the margins takes the full path by itself - results are identical either way.  -> whether this
        plan's shape qualifies."""
        if not (self.front and self.compact and self.C <= 16):
            return False
        if not self.candidates:
            self.candidates = True
            if self.capacity:
                self._alloc_lists(self.capacity)
        self.cand_margins = (float(rot_max), float(trans_max))
        return True

    def _cand_layout(self):
        lay = (C.c_int64 * 4)()
        check(lib.gsx_front_cand_layout(self.N, self.C, self.tile_w, self.tile_h, self.capacity, lay), "gsx_front_cand_layout")
        return int(lay[0]), int(lay[1]), int(lay[2]), int(lay[3])

    def _cand_header(self) -> torch.Tensor:
        hdr_off, _, _, n = self._cand_layout()
        return self.isect_ws[hdr_off:hdr_off + 4 * n].view(torch.float32)

    def build_candidates(self, st: Optional[int] = None):
        """once per frame (per refinement), with ``self.viewmats`` holding the poses the closures start from, and again
        whenever the map's arrays were written: one launch on ``st`` (default: torch's current stream)"""
        if not self.candidates or self.capacity == 0:
            return
        st = current_stream_ptr(self.dev) if st is None else st
        m = self.map
        check(lib.gsx_front_candidates(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C, self.W,
                                       self.H, self.eps2d, self.near, self.far,
                                       self.flags | (_MAP_RECORDS if self.map_records else 0), _p(m[3]), _p(m[4]), _p(m[5]),
                                       self.cand_margins[0], self.cand_margins[1], self.capacity, _p(self.isect_ws),
                                       self.isect_ws.numel(), st), "gsx_front_candidates")

    def candidate_stats(self):
        """(candidates in all rows, mode of the last closure (1 = candidates used), closures that fell back to the full path
        since the last build) - three small blocking reads, for tests and diagnostics"""
        hdr_off, n_off, _, _ = self._cand_layout()
        lay = (C.c_int64 * 4)()
        check(lib.gsx_front_layout(self.N, self.C, self.tile_w, self.tile_h, self.capacity, lay), "gsx_front_layout")
        counts = self.isect_ws[n_off:n_off + 4 * int(lay[0])].view(torch.int32)
        hdr = self._cand_header()
        return int(counts.sum().item()), int(hdr[16 * 12 + 4].item()), int(hdr[16 * 12 + 5].item())

    def probe(self, stream_ptr: Optional[int] = None):
        """sizes the tile lists from the current map and poses with ONE synchronous read of sum(tiles_per_gauss) (an upper
        bound of M that the projection produces anyway); later drift is caught by check_capacity()"""
        st = current_stream_ptr(self.dev) if stream_ptr is None else stream_ptr
        self._project(st, tiles=True)
        if stream_ptr is not None:
            check(lib.gsx_stream_synchronize(st), "gsx_stream_synchronize")
        m = int(self.tiles.sum().item())
        want = int(m * self.GROW) + 4096
        if want > self.capacity:
            self._alloc_lists(want)
        return m

    def keys_total(self) -> int:
        """M of the last render (one blocking read): the front's count, or - row keys - the sum of the 64 key counters"""
        if self.row_keys and self._rows_last:
            return int(self.key_counters.sum().item())
        return int(self.M_dev.item())

    def check_capacity(self) -> bool:
        """after the launches have been issued: True iff no render since the last check overflowed the tile lists.  On
        False the capacity has been grown (``stale`` is set): re-capture and redo.  One blocking read of 12 bytes."""
        st, m = int(self.status.item()), self.keys_total()
        self.last_M = m
        if st & 2:
            # the binning kernels clamped a negative / out-of-range tile count (hardening flag, csrc/isect_bin.hip): the
            # lists of this render were built from corrupt counts - never a silent "ok"
            self.status.zero_()
            raise RuntimeError(f"corrupt tile counts in a render plan (status {st}, M {m}, N {self.N}, C {self.C}, "
                               f"{self.W}x{self.H}, capacity {self.capacity}): the count matrix was overwritten or the "
                               "inputs are not finite")
        if st & 1:
            self.status.zero_()
            self._alloc_lists(int(max(m, self.capacity) * self.GROW) + 4096)
            return False
        return True

    def matches(self, splats) -> bool:
        """the plan still describes this map: same tensors (addresses are baked into captured graphs), same N"""
        ts = [splats.means, splats.quats, splats.scales, splats.opacities, splats.colors, splats.log_uncertainties]
        return all(a.data_ptr() == b.data_ptr() and a.shape == b.shape for a, b in zip(ts, self.map))

    def slot_flatten_ids(self) -> torch.Tensor:
        """compact plans: int64 [n_slots] flatten id (c * N + g) of every instance slot, from the front's instance records
        (valid for the slots below each segment's count; for tests and debugging - one small gather)"""
        assert self.compact
        lay = (C.c_int64 * 4)()
        check(lib.gsx_front_layout(self.N, self.C, self.tile_w, self.tile_h, self.capacity, lay), "gsx_front_layout")
        n_slots = self.C * int(lay[0]) * int(lay[1])
        recs = self.isect_ws[int(lay[2]):int(lay[2]) + n_slots * 16].view(torch.int32).view(n_slots, 4)
        ids = recs[:, 3].long() & 0xFFFFFFFF
        if self.candidates and self.candidate_stats()[1] == 1:
            # the last closure ran on the candidate set: the records name candidate records, whose word 14 is the Gaussian
            _, _, cand_off, _ = self._cand_layout()
            n_cand = int(lay[0]) * int(lay[1])
            cg = self.isect_ws[cand_off:cand_off + n_cand * 64].view(torch.int32).view(n_cand, 16)[:, 14].long()
            cam = (recs[:, 1].long() >> 24) & 0xFF
            ids = cam * self.N + cg[ids.clamp(max=n_cand - 1)]
        return ids

    def as_output(self):
        """the plan's buffers seen as the reference's RasterizationOutput (views, no copies): what pruning, insertion and
        the SYNC payload read after a render (gslam/rasterization.py:17-41).  ``means2d.grad`` is the view of the
        gradient records that ``means2d.retain_grad()`` would have produced (backend.py:326)."""
        from .rasterization import RasterizationOutput
        if self.compact:
            raise RuntimeError("a pose-only plan keeps its records per visible instance: no RasterizationOutput view")
        means2d = self.means2d
        if self.v_rec is not None:
            means2d = self.means2d.view(self.C, self.N, 2)
            means2d.grad = self.v_rec[..., 0:2]
        if not self.front and self.lean_rows:
            # the renders of this plan left the culled rows of means2d / depths / records as they were (lean_rows): the reference
            # has zeros there (gslam/rasterization.py:153-170) - one projection launch with every row written, on demand
            self._project(current_stream_ptr(self.dev), full_rows=True)
        if not self.front:
            # the projection of a render leaves tiles_per_gauss out (16 MB at 500 k x 8 that only this view reads)
            check(lib.gsx_isect_count(_p(self.means2d), _p(self.radii), self.C * self.N, self.tile_w, self.tile_h,
                                      _p(self.tiles), current_stream_ptr(self.dev)), "gsx_isect_count")
        out = RasterizationOutput(
            rgbs=self.render[..., :3], alphas=self.alphas, tile_width=self.tile_w, tile_height=self.tile_h,
            tiles_per_gauss=self.tiles, isect_offsets=self.offsets[:-1].view(self.C, self.tile_h, self.tile_w),
            width=self.W, height=self.H, tile_size=TILE, n_cameras=self.C, camera_ids=None, gaussian_ids=None,
            radii=self.radii, means2d=means2d, depths=self.depths, conics=self.rec[..., 2:5], opacities=self.rec[..., 5],
            n_touched=self.n_touched)
        if self.depth_index >= 0:
            out.depthmaps = self.render[..., self.depth_index]
        out.betas = self.render[..., self.betas_index]
        out._render, out._depth_index, out._betas_index, out._vis_count = self.render, (
            self.depth_index if self.depth_index >= 0 else None), self.betas_index, self.vis_count
        out._plan = self
        return out

    # ---- launches ----------------------------------------------------------------------------------------------------
    def _clear_ptr(self):
        """the gradient records as the forward's clear target - only when an earlier backward(keep=True) left them dirty (an
        eager call decides this when it is issued; owners of captured graphs call clean() before a replay)"""
        if self.v_rec is None:
            return None
        if self.compact:
            # records per visible instance (pose-only plans): 48 dense bytes per instance, cleared on the way by the
            # projection - cheaper there than zeroing them from the pose backward (measured: 12.2 against 11.1 us)
            return self.v_rec.data_ptr()
        if self._v_rec_dirty:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("capture with clean gradient records (RenderPlan.clean())")
            self._v_rec_dirty = False
            return self.v_rec.data_ptr()
        return None

    def clean(self, st: int):
        """zeroes the gradient records if a backward(keep=True) left them behind; call before replaying a captured forward"""
        if self.v_rec is not None and self._v_rec_dirty:
            check(lib.gsx_zero_words(_p(self.v_rec), self.v_rec.numel(), st), "gsx_zero_words")
            self._v_rec_dirty = False

    def _project(self, st: int, tiles: bool = False, full_rows: bool = False):
        """tiles: also write tiles_per_gauss [C,N] (the capacity probe sums it; a render does not need it).  full_rows: write
        the culled rows' zeros even in a plan that leaves them out (lean_rows; as_output() asks for it)"""
        m = self.map
        tight = self.tight_lists or (bool(self.front) and self.tile_exact)
        if (tight or self.rect_lists) and max(self.tile_w, self.tile_h) < 256:
            # the projection packs every instance's tile rectangle (4 bytes; tight under tight_lists) for the binning to read
            # instead of means2d + radius (12 bytes, twice).  (A fused-front plan comes here for its capacity probe only: the
            # count then is that of the tight rectangles its front will list.)
            if self.rects is None:
                self.rects = torch.zeros(self.C, self.N, dtype=torch.int32, device=self.dev)
            check(lib.gsx_project_fwd_rects(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C, self.W,
                                            self.H, self.eps2d, self.near, self.far, 0.0,
                                            self.flags | (_TILE_EXACT if tight else 0) | (
                                                _SKIP_CULLED if self.lean_rows and not full_rows else 0), _p(self.radii),
                                            _p(self.means2d), _p(self.depths), None, None, _p(self.tiles) if tiles else None,
                                            self.tile_w, self.tile_h, _p(m[3]), _p(m[4]), _p(m[5]), _p(self.rec),
                                            _p(self.vis_count), (None if full_rows else self._clear_ptr()), _p(self.rects), st), "gsx_project_fwd_rects")
            return
        check(lib.gsx_project_fwd(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C, self.W,
                                  self.H, self.eps2d, self.near, self.far, 0.0, self.flags, _p(self.radii),
                                  _p(self.means2d), _p(self.depths), None, None, _p(self.tiles) if tiles else None, self.tile_w,
                                  self.tile_h, _p(m[3]), _p(m[4]), _p(m[5]), _p(self.rec), _p(self.vis_count),
                                  (None if full_rows else self._clear_ptr()), st), "gsx_project_fwd")

    def _front(self, st: int, defer_sort: bool = False):
        m = self.map
        lean = self.lean
        rows = bool(defer_sort and self.row_keys)
        self._rows_last = rows
        m_dev = self.key_counters if rows else self.M_dev
        flags = self.flags | (_SKIP_CULLED if lean else 0) | (_COMPACT if self.compact else 0) | self._cand_flags() | (
            _DEFER_SORT if defer_sort else 0) | (_ROW_KEYS if rows else 0) | (_TILE_EXACT if self.tile_exact else 0)
        if defer_sort and self.near_place:
            check(lib.gsx_front_fwd_near(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C, self.W,
                                         self.H, self.eps2d, self.near, self.far, flags, _p(m[3]), _p(m[4]), _p(m[5]),
                                         _p(self.rec), self._clear_ptr(), self.capacity, _p(self.offsets), _p(self.M_dev),
                                         _p(self.status), _p(self.tile_work), _p(self.balanced_order), self.CHUNK_COST,
                                         self.LIGHT_RATE, self.n_cus, _p(self.isect_ws), self.isect_ws.numel(),
                                         _p(self.tile_cut), _p(self.tile_placed), st), "gsx_front_fwd_near")
            return
        check(lib.gsx_front_fwd(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C, self.W,
                                self.H, self.eps2d, self.near, self.far, flags,
                                _p(m[3]), _p(m[4]), _p(m[5]), None if self.compact else _p(self.radii),
                                None if lean else _p(self.means2d), None if lean else _p(self.depths),
                                None, None if self.compact else _p(self.tiles),
                                _p(self.rec), self._clear_ptr(), None if self.compact else _p(self.vis_count),
                                self.capacity, _p(self.offsets),
                                _p(m_dev), _p(self.status), _p(self.flat),
                                None if rows else _p(self.tile_order),
                                _p(self.tile_work), _p(self.balanced_order), self.CHUNK_COST, self.LIGHT_RATE, self.n_cus,
                                _p(self.isect_ws), self.isect_ws.numel(), st), "gsx_front_fwd")

    # next closure's depth cut-off of a tile = depth of its deepest composited entry * (1 + this).  Measured on the headline
    # sequence (tools/dbg/cut_prof.sh; 253 closures x 1200 tiles): 0.02 -> 10 % of the tiles need a second slab, 0.05 -> 3.9 %,
    # 0.1 -> 1.4 %, 0.2 -> 0.4 %, 0.5 -> 0.01 %, while the keys sorted per closure grow from 17 % to 27 % of all; the fused launch
    # reads 81.9 / 81.2 / 80.8 / 81.1 / 81.5 us at 0.05 / 0.1 / 0.2 / 0.3 / 0.5
    CUT_MARGIN = 0.2

    # with the near placement the cut-off also decides which keys are WRITTEN; a tile whose cut-off was too tight reads every instance
    # record of its camera to find the rest (tile_sort_lds.h complete_tile), so the margin is chosen for that to be rare
    NEAR_MARGIN = 0.5

    def enable_near_placement(self, margin: Optional[float] = None) -> bool:
        """On top of ``enable_defer_sort``: the front counts the keys behind a tile's depth cut-off without placing them
        (gsx_front_fwd_near) and the rasteriser's tile workgroups append them themselves in the rare case a pixel outlives the
        placed ones (gsx_raster_track_fused_near).  ``offsets`` / ``M_dev`` still describe the full lists; ``tile_placed[t]`` is what
        the front wrote.  Same consumed lists, entry for entry.  -> whether this plan's shape qualifies."""
        if not self.enable_defer_sort(self.NEAR_MARGIN if margin is None else margin):
            return False
        if not self.near_place:
            self.near_place = True
            self.tile_placed = torch.zeros(self.T, dtype=torch.int32, device=self.dev)
        return True

    def enable_tight_lists(self, on: bool = True) -> bool:
        """plans on the generic chain whose tile lists only their own rasteriser reads (an output render, a BA window): tight
        rectangles packed by the projection (RenderPlan.TIGHT_LISTS says what was measured).  -> whether it applies"""
        ok = bool(on) and not self.front and max(self.tile_w, self.tile_h) < 256
        self.tight_lists = ok
        return ok

    def enable_tile_exact(self, on: bool = True) -> bool:
        """Fused-front plans (GSX_PROJ_TILE_EXACT): the projection lists an instance only in the tiles of its 3-sigma square that hold a
        pixel centre inside the bounding box of its alpha >= 1/255 ellipse - the rasteriser skips the others pixel by pixel anyway
        (render, loss and gradients unchanged; 28 % of the pairs of the headline's map).  -> whether it applies to this plan."""
        ok = bool(on) and bool(self.front)
        self.tile_exact = ok
        return ok

    def enable_row_keys(self) -> bool:
        """On top of ``enable_defer_sort``: the front stops after its projection launch - every projection workgroup leaves its row's
        keys grouped by tile in the row's own segment (GSX_PROJ_ROW_KEYS) - and the rasteriser's tile workgroups collect their keys
        themselves (gsx_raster_track_fused_rows): the column scan and the placement launch are gone from the closure's chain.
        ``offsets`` is not written; ``tile_span[t] = (start, count)`` of every tile's segment and ``M_dev`` come out of the
        rasteriser launch.  Same composited lists, entry for entry.  -> whether this plan's shape qualifies."""
        if not (self.defer_sort and self.compact and not self.near_place):
            return False
        lay = (C.c_int64 * 4)()
        check(lib.gsx_front_rows_layout(self.N, self.C, self.tile_w, self.tile_h, 4096, lay), "gsx_front_rows_layout")
        if int(lay[3]) > 768:
            return False
        if not self.row_keys:
            self.row_keys = True
            self.tile_span = torch.zeros(self.T, 2, dtype=torch.int32, device=self.dev)
            # 64 key counters (tile t draws its segment on counter t % 64; their sum is the render's M)
            self.key_counters = torch.zeros(64, dtype=torch.int64, device=self.dev)
        return True

    def enable_defer_sort(self, margin: Optional[float] = None) -> bool:
        """Pose-only plans on the fused front whose closure runs gsx_raster_track_fused: the front stops after the placement and
        the rasteriser's tile workgroups sort what they composite (the keys up to the tile's depth cut-off of the previous
        closure; the whole segment if a pixel outlives that) - one launch less on the closure's chain, identical lists.  After a
        closure ``self.flat`` holds sorted ids only for the first ``self.tile_near[t]`` entries of tile t.
        -> whether this plan's shape qualifies."""
        if not (self.front and self.compact and self.CH == 4 and self.geom_only):
            return False
        if margin is not None:
            self.cut_margin = float(margin)
        if not self.defer_sort:
            self.defer_sort = True
            self.tile_cut = torch.full((self.T,), 0x7f800000, dtype=torch.int32, device=self.dev)     # +inf: no cut-off yet
            self.tile_near = torch.zeros(self.T, dtype=torch.int32, device=self.dev)
            self.sort_stats = torch.zeros(4, dtype=torch.int32, device=self.dev)
        return True

    def reset_cuts(self):
        """forget the tiles' depth cut-offs (the next closure sorts every tile's whole segment)"""
        if self.tile_cut is not None:
            self.tile_cut.fill_(0x7f800000)

    CHUNK_COST = 3.0        # weight of a 64-entry chunk (gather + cull) in compositing trips (tools/dbg/wg_trace.py)
    LIGHT_RATE = 0.92       # a CU with 4 workgroups gets through 0.92 of the work of one with 5 in the same time (same trace)

    def enable_balance(self) -> bool:
        """-> whether this render's shape qualifies (all workgroups resident at once, several per CU) AND the dispatcher
        places workgroups the way the balanced order assumes (``placement_ok``: probed once per device and shape)"""
        if self.front and self.n_cus and self.n_cus < self.T <= min(5 * self.n_cus, 2048) and self.balanced_order is None:
            ok, why = placement_ok(self.dev, self.T, self.n_cus)
            self.balance_note = why
            if ok:
                self.balanced_order = torch.arange(self.T, dtype=torch.int32, device=self.dev)
                self.tile_work = torch.zeros(self.T, 2, dtype=torch.int32, device=self.dev)
        return self.balanced_order is not None

    def rebalance(self, st: int):
        """new launch order from the work counters the last track-loss forward left, as a launch of its own (the fused front
        does the same inside its projection launch at every forward(): nothing to call in a loop of closures)"""
        if self.balanced_order is not None:
            check(lib.gsx_tile_balance(_p(self.tile_work), self.T, self.CHUNK_COST, self.LIGHT_RATE, self.n_cus,
                                       _p(self.balanced_order), st),
                  "gsx_tile_balance")

    @property
    def launch_order(self):
        return self.balanced_order if self.balanced_order is not None else self.tile_order

    def forward_track_fused(self, st: int, track_loss, refiner_loss: bool = False):
        """front + gsx_raster_track_fused: the forward rasteriser, the tracking loss and the geometry-only rasteriser backward
        of every tile in one launch (pose-only CH = 4 plans); follow with backward(st, rasterised=True)"""
        assert self.CH == 4 and self.geom_only and self.n_touched is None and self.capacity > 0
        assert not refiner_loss or (self.front and self.defer_sort and self.row_keys), "the refiner's loss: row-keys plans only"
        gt, exposure, w_photo, rows = track_loss
        if self.front and self.defer_sort and self.row_keys:
            self._front(st, defer_sort=True)
            check(lib.gsx_raster_track_fused_rows(
                _p(self.rec), _p(self.backgrounds), _p(self.flat), self.capacity, self.N, self.C, self.W, self.H, _p(gt),
                _p(exposure), float(w_photo), 1 if refiner_loss else 0, None, None, None, _p(rows), _p(self.v_rec),
                _p(self.balanced_order),      # (never self.tile_order: a front with row keys does not write the length-sorted order)
                _p(self.tile_work), _p(self.tile_cut), float(self.cut_margin), _p(self.tile_near), _p(self.sort_stats),
                _p(self.tile_span), _p(self.key_counters), _p(self.status), _p(self.isect_ws), self.isect_ws.numel(), st),
                "gsx_raster_track_fused_rows")
            return
        if self.front and self.defer_sort:
            self._front(st, defer_sort=True)
            lay = (C.c_int64 * 3)()
            check(lib.gsx_front_keys(self.N, self.C, self.tile_w, self.tile_h, self.capacity,
                                     _COMPACT if self.compact else 0, lay), "gsx_front_keys")
            base = self.isect_ws.data_ptr()
            if self.near_place:
                l4 = (C.c_int64 * 4)()
                check(lib.gsx_front_layout(self.N, self.C, self.tile_w, self.tile_h, self.capacity, l4), "gsx_front_layout")
                check(lib.gsx_raster_track_fused_near(
                    _p(self.rec), _p(self.backgrounds), _p(self.offsets), _p(self.flat), self.capacity, 1, self.C, self.W, self.H,
                    _p(gt), _p(exposure), float(w_photo), None, None, None, _p(rows), _p(self.v_rec), _p(self.launch_order),
                    _p(self.tile_work), base + int(lay[0]), base + int(lay[1]), int(lay[2]), _p(self.tile_cut),
                    float(self.cut_margin), _p(self.tile_near), _p(self.sort_stats), _p(self.tile_placed), base + int(l4[2]),
                    base + int(l4[3]), int(l4[0]), int(l4[1]), 1 if self.compact else 0, st), "gsx_raster_track_fused_near")
                return
            check(lib.gsx_raster_track_fused_sorting(
                _p(self.rec), _p(self.backgrounds), _p(self.offsets), _p(self.flat), self.capacity, 1, self.C, self.W, self.H,
                _p(gt), _p(exposure), float(w_photo), None, None, None, _p(rows), _p(self.v_rec), _p(self.launch_order),
                _p(self.tile_work), base + int(lay[0]), base + int(lay[1]), int(lay[2]), _p(self.tile_cut),
                float(self.cut_margin), _p(self.tile_near), _p(self.sort_stats), st), "gsx_raster_track_fused_sorting")
            return
        if self.front:
            self._front(st)
        else:
            self._project(st)
            self._isect(st)
        check(lib.gsx_raster_track_fused(_p(self.rec), _p(self.backgrounds), _p(self.offsets), _p(self.flat), self.capacity,
                                         1, self.C, self.W, self.H, _p(gt), _p(exposure), float(w_photo), None, None, None,
                                         _p(rows), _p(self.v_rec), _p(self.launch_order), _p(self.tile_work), st),
              "gsx_raster_track_fused")

    def _isect(self, st: int):
        """tile lists of the generic chain (after ``_project``).  tight_lists: an instance is listed only in the tiles of its 3-sigma
        square that hold a pixel centre inside the box of its alpha >= 1/255 ellipse (gsx_project_fwd_rects | GSX_PROJ_TILE_EXACT ->
        gsx_isect_bin_sort_rects) - what the
        rasteriser composites, and so every output of the render and its backward, is unchanged; ``offsets`` / ``flat`` are then
        NOT gsplat's isect_offsets / flatten_ids (nobody outside gslam/rasterization.py reads those)"""
        if (self.tight_lists or self.rect_lists) and max(self.tile_w, self.tile_h) < 256:      # (as _project decides)
            check(lib.gsx_isect_bin_sort_rects(_p(self.rects), _p(self.depths), self.N, self.C, self.tile_w, self.tile_h,
                                               self.capacity, _p(self.offsets), _p(self.M_dev), _p(self.status), None,
                                               _p(self.flat), _p(self.tile_order), _p(self.isect_ws), self.isect_ws.numel(), st),
                  "gsx_isect_bin_sort_rects")
            return
        check(lib.gsx_isect_bin_sort(_p(self.means2d), _p(self.radii), _p(self.depths), self.N, self.C, self.tile_w,
                                     self.tile_h, self.capacity, _p(self.offsets), _p(self.M_dev), _p(self.status),
                                     None, _p(self.flat), _p(self.tile_order), _p(self.isect_ws),
                                     self.isect_ws.numel(), st), "gsx_isect_bin_sort")

    def forward(self, st: int, track_loss=None):
        """track_loss = (gt [C,H,W,3], exposure [C,2], w_photo, loss_rows [T,6]): the forward rasteriser evaluates the
        active-nerf tracking loss in its epilogue (CH = 4 plans): self.v_render and the loss rows come out of the same
        launch, the render itself is not written"""
        if self.capacity == 0:
            raise RuntimeError("RenderPlan.probe() first (tile-list capacity)")
        if self.front:
            self._front(st)
        else:
            self._project(st)
            self._isect(st)
        if track_loss is not None:
            gt, exposure, w_photo, rows = track_loss
            assert self.CH == 4 and self.n_touched is None
            check(lib.gsx_raster_fwd_track_loss(_p(self.rec), _p(self.backgrounds), _p(self.offsets), _p(self.flat),
                                                self.capacity, 1, self.C, self.W, self.H, _p(gt), _p(exposure),
                                                float(w_photo), None, _p(self.alphas), _p(self.last_ids),
                                                _p(self.v_render), _p(rows), _p(self.launch_order), _p(self.tile_work), st),
                  "gsx_raster_fwd_track_loss")
            return
        if self.n_touched is not None:
            check(lib.gsx_zero_words(_p(self.n_touched), self.n_touched.numel(), st), "gsx_zero_words")
        check(lib.gsx_raster_fwd(_p(self.rec), self.CH, _p(self.backgrounds), _p(self.offsets), _p(self.flat),
                                 self.capacity, 1, self.C, self.W, self.H, self.tile_w, self.tile_h, self.vis_min_T,
                                 _p(self.render), _p(self.alphas), _p(self.last_ids), _p(self.n_touched),
                                 _p(self.tile_order), st), "gsx_raster_fwd")

    def backward(self, st: int, keep: bool = False, rasterised: bool = False, tail=None):
        """from ``self.v_render`` (filled by the loss launch) to the pose partials in ``self.pose_ws`` and, for 'full', the
        six map gradients (overwritten, summed over cameras inside).  keep: leave the gradient records as accumulated
        (``as_output().means2d.grad`` reads them: densification) instead of zeroing each row once it has been consumed.
        rasterised: the gradient records are complete already (forward_track_fused ran the rasteriser's backward)"""
        if not rasterised:
            self.backward_raster(st)
        self.backward_project(st, keep, tail=tail)

    def backward_raster(self, st: int):
        """the rasteriser's half of ``backward``: d loss / d render -> the gradient records of the visible pairs"""
        assert self.grads != 'none'
        check(lib.gsx_raster_bwd(_p(self.rec), self.CH, _p(self.backgrounds), _p(self.offsets), _p(self.flat),
                                 self.capacity, 1, self.C, self.W, self.H, self.tile_w, self.tile_h, _p(self.alphas),
                                 _p(self.last_ids), _p(self.v_render), None, _p(self.v_rec), None,
                                 _p(self.launch_order), 1 if self.geom_only else 0, st), "gsx_raster_bwd")

    def backward_project(self, st: int, keep: bool = False, rows=None, tail=None):
        """the projection's half of ``backward``.  rows = (g_begin, g_end), 'full' plans only: those rows of the map alone
        (gsx_project_bwd_range; g_begin a multiple of 256, g_end a multiple of 256 or N) - calls over ranges that tile the map
        leave the six gradients and the pose partials as one call does"""
        assert self.grads != 'none'
        reset = 0 if (keep or self.compact) else _RESET_V_REC
        if keep and not self.compact:
            self._v_rec_dirty = True
        m = self.map
        vr = self.v_rec.data_ptr()
        if self.grads == 'pose' and self.front:
            flags = self.flags | reset | (_COMPACT if self.compact else 0) | self._cand_flags()
            if tail is not None:
                # the closure's tail inside this launch (its last workgroup runs it): gsx_front_pose_bwd_tail
                state, slots, exposure, loss_rows, n_rows, coef, tickets = tail
                check(lib.gsx_front_pose_bwd_tail(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.W,
                                                  self.H, self.eps2d, self.near, self.far, flags, vr, self.capacity,
                                                  _p(self.isect_ws), self.isect_ws.numel(), _p(self.pose_ws), _p(state),
                                                  _p(slots.Rt), _p(slots.dt), _p(slots.dR), _p(exposure), _p(self.viewmats),
                                                  _p(loss_rows), n_rows, coef, _p(tickets), st), "gsx_front_pose_bwd_tail")
                return
            check(lib.gsx_front_pose_bwd(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C,
                                         self.W, self.H, self.eps2d, self.near, self.far, flags, vr, self.capacity,
                                         _p(self.isect_ws), self.isect_ws.numel(), _p(self.pose_ws), st),
                  "gsx_front_pose_bwd")
            return
        assert tail is None
        if self.grads == 'pose':
            outs = (None,) * 3 + (None,) + (None,) * 3
        else:
            v = self.v_map
            outs = (_p(v[0]), _p(v[1]), _p(v[2]), None, _p(v[3]), _p(v[4]), _p(v[5]))
        if rows is not None:
            assert self.grads == 'full'
            check(lib.gsx_project_bwd_range(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C,
                                            self.W, self.H, self.eps2d, self.near, self.far,
                                            self.flags | _VIEW_PARTIALS | reset, _p(self.radii), vr, 12, None, vr + 8, 12,
                                            None, _p(m[3]), _p(m[4]), _p(m[5]), vr, *outs, _p(self.pose_ws),
                                            self.pose_ws.numel(), int(rows[0]), int(rows[1]), st), "gsx_project_bwd_range")
            return
        check(lib.gsx_project_bwd(_p(m[0]), _p(m[1]), _p(m[2]), _p(self.viewmats), _p(self.Ks), self.N, self.C, self.W,
                                  self.H, self.eps2d, self.near, self.far, self.flags | _VIEW_PARTIALS | reset, _p(self.radii),
                                  vr, 12, None, vr + 8, 12, None, _p(m[3]), _p(m[4]), _p(m[5]), vr, *outs,
                                  _p(self.pose_ws), self.pose_ws.numel(), st), "gsx_project_bwd")


class _PoseSlots:
    """PoseZhou parameters of up to 16 cameras in three persistent arrays (rows = poses), with the host pointer tables
    the pose launches take (gslam/primitives.py:40-92)."""

    def __init__(self, n: int, dev, learnable: Sequence[bool]):
        self.n = n
        self.Rt = torch.eye(4, device=dev).repeat(n, 1, 1).contiguous()
        self.dR = torch.zeros(n, 6, device=dev)
        self.dt = torch.zeros(n, 3, device=dev)
        self.v_dR = torch.zeros(n, 6, device=dev)
        self.v_dt = torch.zeros(n, 3, device=dev)
        self.learnable = [bool(x) for x in learnable]
        self._flags = (C.c_int * n)(*[1 if x else 0 for x in self.learnable])
        rows = lambda t: _arr([t[i].data_ptr() for i in range(n)])
        self._Rt, self._dR, self._dt, self._vdR, self._vdt = rows(self.Rt), rows(self.dR), rows(self.dt), rows(self.v_dR), rows(self.v_dt)

    def forward(self, viewmats: torch.Tensor, st: int):
        check(lib.gsx_pose_zhou_fwd(self.n, self._Rt, self._dR, self._dt, self._flags, viewmats.data_ptr(), st),
              "gsx_pose_zhou_fwd")

    def backward_partials(self, r: RenderPlan, st: int):
        if not any(self.learnable):
            return
        check(lib.gsx_pose_zhou_bwd_partials(self.n, self._Rt, self._dR, self._dt, self._flags, _p(r.pose_ws),
                                             r.pose_blocks, None, self._vdR, self._vdt, st),
              "gsx_pose_zhou_bwd_partials")

    @torch.no_grad()
    def load(self, poses):
        for i, p in enumerate(poses):
            self.Rt[i].copy_(p.Rt)
            if self.learnable[i]:
                self.dR[i].copy_(p.dR)
                self.dt[i].copy_(p.dt)
            else:
                self.dR[i].zero_()
                self.dt[i].zero_()

    @torch.no_grad()
    def store(self, poses):
        for i, p in enumerate(poses):
            if self.learnable[i]:
                p.dR.copy_(self.dR[i])
                p.dt.copy_(self.dt[i])


def _photometric_loss(r: RenderPlan, gt: torch.Tensor, exposure: torch.Tensor, mode: int, w_photo: float, map_ws, st: int,
                      use_alphas: bool):
    """csrc/loss.hip map_loss_kernel: per-pixel value partials (left in ``map_ws``) and d loss / d render -> r.v_render"""
    denom = r.C * r.H * r.W * (3 if mode == 1 else 1)
    check(lib.gsx_map_loss(_p(r.render), _p(r.alphas) if use_alphas else None, _p(gt), _p(exposure), r.C, r.H, r.W, r.CH,
                           r.depth_index, r.betas_index, mode, w_photo / denom, 0.0, 0.4, None, None, _p(r.v_render),
                           None, _p(map_ws), map_ws.numel(), st), "gsx_map_loss")
    return denom


class TrackClosure:
    """One evaluation of the tracking closure (gslam/frontend.py:621-649) on a persistent slot - pose (Rt, dR, dt),
    exposure [2], target image - against a frozen map: C = 1 render without the depth channel (the reference renders it
    but reads it only under use_gt_depths, frontend.py:134-137), active-nerf loss with the exposure affine (:113-138,
    :632-636), backward to the 9 pose scalars + 2 exposure scalars.

    tail = 'fused': ONE launch takes the projection backward's pose partials through the PoseZhou backward, finishes the
    loss, advances the device optimiser (10 Adam steps + strong-Wolfe L-BFGS, csrc/track_opt.h) and writes the next
    evaluation point and its view matrix.  'split': the same with separate launches (PoseZhou forward, partials ->
    PoseZhou backward, loss finish, optimiser advance).  'host': stops at the gradients (``g_dt, g_dR, g_exposure``,
    ``loss``) for an optimiser on the host."""

    # default of ``merge_tail``.  OFF: built, parity-green, measured SLOWER (prof_closure, 253 closures, same box): the pose backward
    # with the tail inside 22.9-23.1 us against 10.6 + 10.0 us for the two launches.  With release / acquire fences around the ticket
    # it was 34 us (every workgroup's fence writes back and invalidates its XCD's L2 under the workgroups still at work); without
    # them - write-through stores, an acknowledged-store wait, relaxed tickets, agent-scope loads - the last workgroup still pays two
    # dependent atomic round trips to memory and reads 490 rows past every cache: ~12 us for what a launch of its own does in 10
    # including its launch.
    MERGE_TAIL = False
    CAND_MARGINS = (0.02, 0.02)    # |R R0^T - I|_F (~0.8 degrees) and metres the closures of a frame may move from its first pose

    def __init__(self, splats, camera, tail: str = 'fused', fuse_raster: bool = True, front: Optional[bool] = None,
                 candidates: bool = False, defer_sort: Optional[bool] = None, map_records: Optional[bool] = None,
                 near_place: Optional[bool] = None, row_keys: Optional[bool] = None, tile_exact: Optional[bool] = None,
                 merge_tail: Optional[bool] = None):
        """near_place: the front leaves the keys behind a tile's depth cut-off out of the placement (RenderPlan.enable_near_placement).
        OFF by default - built, exact and measured in round 5: with 31 % of the keys written the placement launch is
        as long as before (26.5 against 25.1 us: it is a chain of latencies, not of stores), the projection pays 1.5 us for the
        per-pair cut-off test, and the tiles whose cut-off fails (0.33 per closure at margin 0.5) read every instance record of
        the camera: the fused launch 88.3 against 81.0 us.
        fuse_raster (fused tail only): forward rasteriser, loss and rasteriser backward as ONE launch
        (gsx_raster_track_fused); False keeps them as two (the independent path the tests compare against).
        candidates: per-frame candidate set for the closures' projection (gsx_front_candidates).  OFF by default - measured
        on the headline sequence: the 36 evaluation points of a frame spread over up to ~0.1 rad / 0.1 m (Adam's
        lr-sized steps, the line search's trial points); margins that cover them make the candidate set as large as what
        the per-closure cull keeps (334 k of 500 k at 0.06 / 0.06 against ~170 k visible), and tighter ones send most
        closures to the full path.  Results are identical either way.
        map_records: the pose-independent record of EVERY Gaussian, rebuilt once per frame (RenderPlan.enable_map_records):
        the closures keep their own cull and read one 64-byte record per survivor; None = where it applies (fused front,
        compact records) unless ``candidates`` was asked for.  Results are identical either way.
        defer_sort: the tile sort moves into the fused rasteriser launch (RenderPlan.enable_defer_sort); None = where it
        applies (fused tail + fused rasteriser + fused front)."""
        assert tail in ('fused', 'split', 'host')
        self.tail = tail
        self.fuse_raster = bool(fuse_raster) and tail == 'fused'
        self.camera = camera
        self.r = RenderPlan(splats, 1, camera.width, camera.height, render_depth=False, grads='pose',
                            Ks=camera.intrinsics, front=front)
        # tile_exact: the fused front leaves out the tiles of an instance's 3-sigma square that none of its visible pixels lies in
        # (RenderPlan.enable_tile_exact); None = wherever the fused front runs.  Results are identical either way.
        if tile_exact is None or tile_exact:
            self.r.enable_tile_exact()
            self.r.tight_lists = not self.r.front
        dev = self.r.dev
        self.dev = dev
        self.slots = _PoseSlots(1, dev, [True])
        self.exposure = torch.zeros(2, device=dev)
        self.g_exposure = torch.zeros(2, device=dev)
        self.img = torch.zeros(camera.height, camera.width, 3, device=dev)
        self.out2 = torch.zeros(2, device=dev)
        self._zeros9 = torch.zeros(9, device=dev)
        self.state = torch.zeros(int(lib.gsx_track_opt_state_bytes()), dtype=torch.uint8, device=dev)
        self.report = torch.zeros(8, dtype=torch.float32, device=dev)
        self.map_ws = torch.empty(int(lib.gsx_map_loss_workspace_bytes(1, self.r.H, self.r.W)), dtype=torch.uint8,
                                  device=dev)
        self.n_rows = (self.r.H * self.r.W + 255) // 256
        # fused tail: the loss is evaluated in the forward rasteriser's epilogue, one row of partials per tile
        self.loss_rows = torch.zeros(self.r.T, 6, device=dev)
        # merge_tail: the closure's tail is run by the last workgroup of the pose backward launch (gsx_front_pose_bwd_tail); None =
        # where it applies (fused tail + fused rasteriser + fused front); False keeps gsx_track_opt_tail as its own launch
        self.merge_tail = bool(self.MERGE_TAIL if merge_tail is None else merge_tail) and self.fuse_raster and bool(self.r.front)
        self.tail_tickets = torch.zeros(int(lib.gsx_front_pose_bwd_tail_words()), dtype=torch.int32, device=dev)
        if tail == 'fused':
            self.r.enable_balance()       # the fused forward leaves the tiles' work counters: CU-balanced launch order
        # the 36 closures of a frame re-project the same map for poses that differ by fractions of a pixel: the map is culled
        # ONCE per frame, with margins (gsx_front_candidates), and the closures read that candidate set
        if candidates:
            self.r.enable_candidates(*self.CAND_MARGINS)
        elif map_records is None or map_records:
            self.r.enable_map_records()
        if (defer_sort is None or defer_sort) and self.fuse_raster:
            if near_place:
                self.r.enable_near_placement()
            else:
                self.r.enable_defer_sort()
                # row_keys: no column scan, no placement launch - the rasteriser's tiles collect their keys from the projection
                # rows (RenderPlan.enable_row_keys); None = wherever the sort runs inside the rasteriser
                if row_keys is None or row_keys:
                    self.r.enable_row_keys()
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = HipGraph()
        self._chains: Dict[int, HipGraph] = {}

    def launch(self, count: int, stream_ptr: Optional[int] = None):
        """``count`` closures, one graph launch"""
        _launch_chain(self, count, stream_ptr)

    def rebalance(self):
        """once per frame, before its closures: launch order of the rasteriser kernels from the last closure's tile work"""
        self.r.rebalance(current_stream_ptr(self.dev))

    # convenient views of the slot
    @property
    def Rt(self): return self.slots.Rt[0]
    @property
    def dR(self): return self.slots.dR[0]
    @property
    def dt(self): return self.slots.dt[0]
    @property
    def g_dR(self): return self.slots.v_dR[0]
    @property
    def g_dt(self): return self.slots.v_dt[0]
    @property
    def loss(self): return self.out2[0:1]

    @torch.no_grad()
    def load(self, Rt: torch.Tensor, img: torch.Tensor, exposure: torch.Tensor):
        self.slots.Rt[0].copy_(Rt)
        self.slots.dR.zero_()
        self.slots.dt.zero_()
        self.img.copy_(img)
        self.exposure.copy_(exposure.reshape(2))
        self.r.viewmats[0].copy_(Rt)                    # dR = dt = 0: the first evaluation's view matrix
        self.r.build_candidates()                       # the frame's candidate set, around that pose (map as of now)

    def load_frame(self, pose, img: torch.Tensor, exposure: torch.Tensor):
        """``load`` of a frame whose pose is a PoseZhou module, in THREE launches on torch's current stream instead of the
        ~20 torch ops of ``load(pose().detach(), ...)`` (PoseZhou.forward alone is a normalize / cross / stack / cat chain and a
        4x4 matmul that lands on hipBLASLt): gsx_pose_zhou_fwd writes the frame's view matrix straight into the plan, one
        row-copy launch fills the slot (Rt <- that matrix, dR = dt = 0, exposure, image), then the frame's candidate set."""
        from .transport import copy_rows
        st = current_stream_ptr(self.dev)
        learn = 1 if getattr(pose, "is_learnable", True) else 0
        check(lib.gsx_pose_zhou_fwd(1, _arr([pose.Rt.data_ptr()]), _arr([pose.dR.data_ptr()]), _arr([pose.dt.data_ptr()]),
                                    (C.c_int * 1)(learn), _p(self.r.viewmats), st), "gsx_pose_zhou_fwd")
        z = self._zeros9
        img, exposure = img.detach(), exposure.detach().reshape(1, 2)
        if not (img.is_contiguous() and img.dtype == torch.float32 and img.shape == self.img.shape):
            raise RuntimeError("frame image must be a contiguous float32 [H,W,3] tensor of the plan's size")
        copy_rows([self.slots.Rt.view(1, 16), self.slots.dR.view(1, 6), self.slots.dt.view(1, 3), self.exposure.view(1, 2),
                   self.img.view(1, -1)],
                  [self.r.viewmats.view(1, 16), z[:6].view(1, 6), z[6:].view(1, 3), exposure.contiguous(),
                   img.view(1, -1)], 1)
        self.r.build_candidates(st)

    def store_frame(self, pose, exposure_out: torch.Tensor):
        """the tracked pose and exposure back into the frame's objects, one launch: pose.Rt <- the view matrix the closure's
        tail left in the plan, pose.dR = pose.dt = 0 (frontend.py:659-662 keeps the optimised pose as the frame's pose)"""
        from .transport import copy_rows
        z = self._zeros9
        copy_rows([pose.Rt.view(1, 16), pose.dR.data.view(1, 6), pose.dt.data.view(1, 3), exposure_out.data.view(1, 2)],
                  [self.r.viewmats.view(1, 16), z[:6].view(1, 6), z[6:].view(1, 3), self.exposure.view(1, 2)], 1)

    def enqueue(self, st: int):
        r = self.r
        if self.tail == 'fused':
            denom = r.C * r.H * r.W
            if self.fuse_raster and self.merge_tail:
                # the tail runs inside the pose backward launch (its last workgroup): three launches per closure
                r.forward_track_fused(st, (self.img, self.exposure, 1.0 / denom, self.loss_rows))
                r.backward(st, rasterised=True, tail=(self.state, self.slots, self.exposure, self.loss_rows, r.T, 1.0 / denom,
                                                      self.tail_tickets))
                return
            if self.fuse_raster:
                r.forward_track_fused(st, (self.img, self.exposure, 1.0 / denom, self.loss_rows))
                r.backward(st, rasterised=True)
            else:
                r.forward(st, track_loss=(self.img, self.exposure, 1.0 / denom, self.loss_rows))
                r.backward(st)
            check(lib.gsx_track_opt_tail(_p(self.state), _p(r.pose_ws), r.pose_blocks, _p(self.slots.Rt), _p(self.slots.dt),
                                         _p(self.slots.dR), _p(self.exposure), None, None, _p(r.viewmats),
                                         _p(self.loss_rows), r.T, 1.0 / denom, st), "gsx_track_opt_tail")
            return
        self.slots.forward(r.viewmats, st)
        r.forward(st)
        denom = _photometric_loss(r, self.img, self.exposure, 2, 1.0, self.map_ws, st, use_alphas=False)
        r.backward(st)
        self.slots.backward_partials(r, st)
        pm = 1.0 / denom
        c0 = (C.c_float * 5)(pm, pm, 0.0, 0.0, 0.0)
        check(lib.gsx_loss_finish(_p(self.map_ws), 1, r.H, r.W, None, 0, None, 0, c0, c0, 0.0, 0.0, None,
                                  _p(self.g_exposure), _p(self.out2), st), "gsx_loss_finish")
        if self.tail == 'split':
            ps = [self.dt, self.dR, self.exposure]
            gs = [self.g_dt, self.g_dR, self.g_exposure]
            check(lib.gsx_track_opt_advance(_p(self.state), 3, _arr([t.data_ptr() for t in ps]),
                                            _arr([t.data_ptr() for t in gs]), (C.c_int * 3)(3, 6, 2), _p(self.out2), st),
                  "gsx_track_opt_advance")

    def init_optimizer(self, n_adam: int, lr: float, history: int, max_eval: int, st: Optional[int] = None):
        """torch.optim.LBFGS defaults of the reference call (frontend.py:613-619): max_iter 20, tolerance_grad 1e-7,
        tolerance_change 1e-9"""
        check(lib.gsx_track_opt_init(_p(self.state), 11, n_adam, lr, lr, history, 20, max_eval, 1e-7, 1e-9,
                                     current_stream_ptr(self.dev) if st is None else st), "gsx_track_opt_init")

    def read_report(self, st: Optional[int] = None) -> torch.Tensor:
        check(lib.gsx_track_opt_report(_p(self.state), _p(self.report), current_stream_ptr(self.dev) if st is None else st),
              "gsx_track_opt_report")
        return self.report

    def prepare(self):
        """capacity probe + two eager evaluations + capture.  The slot must hold a frame (load()); the optimiser state is
        whatever it was - callers re-initialise it before the replays that count."""
        saved = [t.clone() for t in (self.slots.dR, self.slots.dt, self.exposure, self.r.viewmats)]
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        st = self.stream.cuda_stream
        for _ in range(4):
            self.r.probe(st)
            self.init_optimizer(10, 0.0, 5, 25, st)          # the warm-up evaluation's step is undone below
            self.enqueue(st)
            self.stream.synchronize()
            if self.r.check_capacity():
                break
        else:
            raise RuntimeError("tile-list capacity keeps overflowing during warm-up")
        _drop_chains(self)
        self.graph.capture(self.stream, self.enqueue)
        self.r.stale = False
        self.stream.synchronize()
        with torch.no_grad():
            for t, s in zip((self.slots.dR, self.slots.dt, self.exposure, self.r.viewmats), saved):
                t.copy_(s)
        torch.cuda.current_stream(self.dev).wait_stream(self.stream)


class WindowClosure:
    """One evaluation of the closure of ``Backend.optimize_poses_lbfgs`` (gslam/backend.py:465-504): the C <= 8 window
    keyframes rendered against the frozen map without the depth channel, photometric term only (:484-492), backward to the
    pose parameters; the strong-Wolfe L-BFGS state machine (history 10, <= 80 parameters, csrc/window_opt.hip) advanced
    at its end.  Poses, exposure and images live in slots of the plan; ``load`` / ``store`` move a window in and out."""

    def __init__(self, splats, cameras, learnable: Sequence[bool], active_gs: bool = True, fused: Optional[bool] = None):
        """fused (round 5): the closure on the tracking closure's machinery - fused front with records per
        visible instance, row keys (no column scan, no placement launch), the tile sort inside the rasteriser up to the tile's
        depth cut-off, forward + the refiner's loss + geometry-only backward of a tile in one launch
        (gsx_raster_track_fused_rows, loss_kind 1), pose gradient over the visible instances: 6 launches for 12.  Parity-green
        (tests/test_gpu_plans.py::test_window_closure_on_the_tracking_machinery_equals_the_generic_path) and, measured at
        500 k x 8 cameras (tools/dbg/window_closure_run.py), SLOWER than the generic chain: 904.6 against 860.1 us per closure -
        the projection workgroups walk 8 cameras one after the other (258 us against 57 + 64 for projection + placement) and the
        9600 tiles collect their keys from 245 rows each (2.35 M stretches of ~6 keys; the fused launch 571 us against 139 + 76 +
        308 for forward + sort + backward).  So the default is the generic chain; True asks for this one where the shape
        qualifies (active_gs loss, the window's tiles fit the front's LDS plan)."""
        Cn = len(cameras)
        self.C = Cn
        Ks = torch.stack([c.intrinsics for c in cameras], dim=0)
        self.fused = False
        if active_gs and fused:
            r = RenderPlan(splats, Cn, cameras[0].width, cameras[0].height, render_depth=False, grads='pose', front=True)
            if r.front and r.compact and r.enable_map_records() and r.enable_defer_sort() and r.enable_row_keys():
                r.enable_tile_exact()
                self.r = r
                self.fused = True
                self.loss_rows = torch.zeros(r.T, 6, device=r.dev)
        if not self.fused:
            self.r = RenderPlan(splats, Cn, cameras[0].width, cameras[0].height, render_depth=False, grads='pose')
            self.r.tight_lists = bool(RenderPlan.TIGHT_LISTS) and not self.r.front and max(self.r.tile_w, self.r.tile_h) < 256
            self.r.rect_lists = bool(RenderPlan.RECT_LISTS) and not self.r.front and max(self.r.tile_w, self.r.tile_h) < 256
            self.r.lean_rows = bool(RenderPlan.LEAN_ROWS) and (self.r.tight_lists or self.r.rect_lists)
            if self.r.front and RenderPlan.TIGHT_LISTS:
                self.r.enable_tile_exact()
        self.r.Ks.copy_(Ks)
        dev = self.r.dev
        self.dev = dev
        self.mode = 0 if active_gs else 1
        self.slots = _PoseSlots(Cn, dev, learnable)
        self.exposure = torch.zeros(Cn, 2, device=dev)
        self.g_exposure = torch.zeros(Cn, 2, device=dev)
        self.gt = torch.zeros(Cn, self.r.H, self.r.W, 3, device=dev)
        self.out2 = torch.zeros(2, device=dev)
        self.state = torch.zeros(int(lib.gsx_window_opt_state_bytes()), dtype=torch.uint8, device=dev)
        self.report = torch.zeros(8, dtype=torch.float32, device=dev)
        self.map_ws = torch.empty(int(lib.gsx_map_loss_workspace_bytes(Cn, self.r.H, self.r.W)), dtype=torch.uint8,
                                  device=dev)
        # parameter order of the host version: for each learnable pose dt, dR (PoseZhou.parameters())
        ps, gs, ns = [], [], []
        for i, l in enumerate(self.slots.learnable):
            if l:
                ps += [self.slots.dt[i], self.slots.dR[i]]
                gs += [self.slots.v_dt[i], self.slots.v_dR[i]]
                ns += [3, 6]
        self.n_tensors, self.n_params = len(ps), sum(ns)
        if self.n_params == 0:
            raise ValueError("no learnable pose in the window")
        if self.n_params > 80 or self.n_tensors > 16:
            raise ValueError("window too large for the device optimiser (80 parameters in 16 tensors)")
        self._ps, self._gs = _arr([t.data_ptr() for t in ps]), _arr([t.data_ptr() for t in gs])
        self._ns = (C.c_int * len(ns))(*ns)
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = HipGraph()
        self._chains: Dict[int, HipGraph] = {}

    @torch.no_grad()
    def load(self, window):
        self.slots.load([x.pose for x in window])
        for i, x in enumerate(window):
            self.gt[i].copy_(x.img)
            self.exposure[i].copy_(x.exposure_params.detach().reshape(2))
        if self.fused:
            self.r.build_candidates()                   # records of the map as it is now (one launch per refinement)

    def store(self, window):
        self.slots.store([x.pose for x in window])

    def enqueue(self, st: int, advance: bool = True):
        r = self.r
        self.slots.forward(r.viewmats, st)
        if self.fused:
            denom = r.C * r.H * r.W
            r.forward_track_fused(st, (self.gt, self.exposure, 1.0 / denom, self.loss_rows), refiner_loss=True)
            r.backward(st, rasterised=True)
            self.slots.backward_partials(r, st)
            pm = 1.0 / denom
            c0 = (C.c_float * 5)(pm, pm, 0.0, 0.0, 0.0)
            # the loss rows are one per TILE here, camera-major: gsx_loss_finish takes "rows per camera = ceil(H * W / 256)", so it is
            # told an image of (tiles per camera) x 256 pixels
            check(lib.gsx_loss_finish(_p(self.loss_rows), r.C, r.tile_w * r.tile_h, 256, None, 0, None, 0, c0, c0, 0.0, 0.0,
                                      None, _p(self.g_exposure), _p(self.out2), st), "gsx_loss_finish")
        else:
            r.forward(st)
            denom = _photometric_loss(r, self.gt, self.exposure, self.mode, 1.0, self.map_ws, st, use_alphas=True)
            r.backward(st)
            self.slots.backward_partials(r, st)
            pm = 1.0 / denom
            c0 = (C.c_float * 5)(pm, pm, 0.0, 0.0, 0.0)
            check(lib.gsx_loss_finish(_p(self.map_ws), r.C, r.H, r.W, None, 0, None, 0, c0, c0, 0.0, 0.0, None,
                                      _p(self.g_exposure), _p(self.out2), st), "gsx_loss_finish")
        if advance:
            check(lib.gsx_window_opt_advance(_p(self.state), self.n_tensors, self._ps, self._gs, self._ns, _p(self.out2),
                                             st), "gsx_window_opt_advance")

    def init_optimizer(self, max_eval: int, st: Optional[int] = None):
        """torch.optim.LBFGS defaults of the reference call (backend.py:465-470): lr 1, max_iter 20, history 10,
        tolerance_grad 1e-7, tolerance_change 1e-7"""
        st = current_stream_ptr(self.dev) if st is None else st
        check(lib.gsx_window_opt_init(_p(self.state), self.n_params, 0, 0.0, 1.0, 10, 20, int(max_eval), 1e-7, 1e-7, st),
              "gsx_window_opt_init")

    def read_report(self, st: Optional[int] = None) -> torch.Tensor:
        check(lib.gsx_window_opt_report(_p(self.state), _p(self.report),
                                        current_stream_ptr(self.dev) if st is None else st), "gsx_window_opt_report")
        return self.report

    def launch(self, count: int, stream_ptr: Optional[int] = None):
        """``count`` closures, one graph launch"""
        _launch_chain(self, count, stream_ptr)

    def prepare(self):
        saved = [t.clone() for t in (self.slots.dR, self.slots.dt)]
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        st = self.stream.cuda_stream
        for _ in range(4):
            self.slots.forward(self.r.viewmats, st)
            self.r.probe(st)
            self.init_optimizer(25, st)                      # the warm-up evaluation's step is undone below
            self.enqueue(st)
            self.stream.synchronize()
            if self.r.check_capacity():
                break
        else:
            raise RuntimeError("tile-list capacity keeps overflowing during warm-up")
        if self.fused:
            self.r.build_candidates(st)                      # (the probe re-allocated the workspace the records live in)
        _drop_chains(self)
        self.graph.capture(self.stream, self.enqueue)
        self.r.stale = False
        self.stream.synchronize()
        with torch.no_grad():
            for t, s in zip((self.slots.dR, self.slots.dt), saved):
                t.copy_(s)
        torch.cuda.current_stream(self.dev).wait_stream(self.stream)


GRAD_PARAMS = ('means', 'quats', 'scales', 'opacities', 'colors', 'log_uncertainties')


class MappingStep:
    """One bundle-adjustment iteration of ``Backend.optimize_map`` (gslam/backend.py:260-359) over a FIXED window as a
    launch plan: PoseZhou forward of the window poses -> render (RGB + depth + beta) -> fused SSIM forward / backward ->
    loss block (exposure affine, active-nerf photometric + 0.5 log^2 beta, edge-aware depth TV, SSIM gradient folded in;
    backend.py:273-318) -> rasteriser + projection backward to all six map arrays and the pose partials -> PoseZhou
    backward -> isotropic term added into the scale gradient -> [all-reduce] -> six splat Adams + pose Adam + opacity
    decay in one launch (backend.py:554-602, :356-359).

    The map gradients live in ONE flat fp32 bucket (tensor-major, 15 N floats), the small things every rank needs whole
    - visible-camera counts [N], pose gradients of all Cw window cameras, loss values, overflow flag - in the "head".
    Multi-GPU (gslam_amd.dist.StepBucket): the window's cameras are dealt round-robin over ranks, each rank
    renders its own against its replica of the map; the head is all-reduced, the gradient bucket REDUCE-SCATTERED (rank r
    gets the window-wide sum of chunk r), each rank runs Adam on its 1 / G of the map (parameters and moments are flat
    buffers of the same layout) and the updated parameter chunks are ALL-GATHERED; every rank applies the identical update
    to all window poses from the head, so replicas never diverge and no pose broadcast is needed.  Per-camera means are
    scaled by C_local / C_window, the TV sum is not, the isotropic term is added once (rank 0) between the two collectives
    from the window-wide visibility.  The step is two graphs around the collectives; with one rank, one graph.

    ``optimizers``: mapping.MapOptimizers built with capturable=True.  Exposure parameters of keyframes are constants
    here (the backend freezes them when it adds a keyframe, gslam_amd/backend.py add_keyframe)."""

    FUSE_SSIM_LOSS = True       # SSIM backward + loss block in one launch (gsx_ssim_bwd_map_loss); False: two launches (A/B, tests)

    def __init__(self, splats, optimizers, window, conf, regularize: bool = True, shard=None,
                 need_n_touched: bool = False, decay_opacity: bool = True, exchange_ranges: int = 0,
                 exchange_overlap: bool = True):
        """exchange_ranges = K > 0 (more than one rank): the RANGED exchange (gslam_amd.dist.StepBucket) - the projection backward
        runs range by range and the reduce-scatter of a finished range is issued on a second stream while the next range is
        computed; the update and the all-gather of the parameters likewise.  exchange_overlap = False keeps the ranged layout
        and issues everything on one stream (the A/B of the overlap itself).  OFF by default: no multi-GPU node has been
        available to measure it on; results are identical to the one-shot exchange (tests/test_gpu_multirank.py)."""
        self.splats, self.optimizers, self.conf = splats, optimizers, conf
        self.window = list(window)
        self.regularize = bool(regularize)
        self.rank = 0 if shard is None else shard.rank
        self.world = 1 if shard is None else shard.world_size
        self.group = None if shard is None else shard.group
        Cw = len(self.window)
        if Cw == 0 or Cw > 16:
            raise ValueError("window of 1..16 keyframes")
        self.Cw = Cw
        self.mine = [i for i in range(Cw) if i % self.world == self.rank]
        if any(f.exposure_params is not None and f.exposure_params.requires_grad for f in self.window):
            raise NotImplementedError("trainable keyframe exposure is not part of the launch plan")
        params = [getattr(splats, n) for n in GRAD_PARAMS]
        N = int(params[0].shape[0])
        dev = params[0].device
        self.dev, self.N = dev, N
        # ---- the bucket (gslam_amd.dist.StepBucket: map gradients | counts | pose rows | loss slots) -------------------
        from .dist import StepBucket
        self.bucket = StepBucket([p.shape for p in params], Cw, dev, group=self.group, world=self.world, rank=self.rank,
                                 ranges=int(exchange_ranges))
        self.ranged = self.bucket.ranges > 0
        self.overlap = bool(exchange_overlap) and self.ranged
        self.flat = self.bucket.flat
        self.grad_views = dict(zip(GRAD_PARAMS, self.bucket.views))
        self.counts, self.g_dt, self.g_dR = self.bucket.counts, self.bucket.g_dt, self.bucket.g_dR
        self.out2, self.vis_i32 = self.bucket.out2, self.bucket.vis_i32
        self.out4, self.overflow = self.bucket.out4, self.bucket.overflow
        # sharded update (world > 1): parameters and Adam moments re-homed into flat buffers of the bucket's layout, BEFORE
        # the render plan below takes their addresses
        self.flat_state = None
        if self.world > 1:
            from .dist import flatten_map_state
            self.flat_state = flatten_map_state(splats, optimizers.splat_opt, self.bucket)
        # ---- poses -----------------------------------------------------------------------------------------------------
        self.learnable = [bool(getattr(f.pose, "is_learnable", True) and f.pose.dR.requires_grad) for f in self.window]
        for f in self.window:
            optimizers.add_pose(f.pose)
            for t in (f.pose.Rt, f.pose.dR, f.pose.dt):
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    raise RuntimeError("poses must be contiguous float32 on the GPU")
        Cl = len(self.mine)
        self.r: Optional[RenderPlan] = None
        cam0 = self.window[0].camera
        self.H, self.W = int(cam0.height), int(cam0.width)
        if Cl > 0:
            self.r = RenderPlan(splats, Cl, self.W, self.H, render_depth=True, grads='full', grad_out=self.grad_views,
                                need_n_touched=need_n_touched)
            self.r.tight_lists = bool(RenderPlan.TIGHT_LISTS) and not self.r.front and max(self.r.tile_w, self.r.tile_h) < 256
            self.r.rect_lists = bool(RenderPlan.RECT_LISTS) and not self.r.front and max(self.r.tile_w, self.r.tile_h) < 256
            self.r.lean_rows = bool(RenderPlan.LEAN_ROWS) and (self.r.tight_lists or self.r.rect_lists)
            if self.r.front and RenderPlan.TIGHT_LISTS:
                self.r.enable_tile_exact()
            self.r.Ks.copy_(torch.stack([self.window[i].camera.intrinsics for i in self.mine], dim=0))
            r = self.r
            self.gt = torch.empty(Cl, self.H, self.W, 3, device=dev)
            self.exposure = torch.zeros(Cl, 2, device=dev)
            self.g_exposure = torch.zeros(Cl, 2, device=dev)
            self.refresh_inputs()
            self._flags = (C.c_int * Cl)(*[1 if self.learnable[i] else 0 for i in self.mine])
            self._Rt = _arr([self.window[i].pose.Rt.data_ptr() for i in self.mine])
            self._dR = _arr([self.window[i].pose.dR.data_ptr() for i in self.mine])
            self._dt = _arr([self.window[i].pose.dt.data_ptr() for i in self.mine])
            self._vdR = _arr([self.g_dR[i].data_ptr() for i in self.mine])
            self._vdt = _arr([self.g_dt[i].data_ptr() for i in self.mine])
            H, W = self.H, self.W
            self.dm = torch.empty(3, Cl, 3, H, W, device=dev)
            self.ssim_grad = torch.empty(Cl, 3, H, W, device=dev)
            self.ssim_ws = torch.empty(int(lib.gsx_ssim_workspace_bytes(Cl, 3, H, W)), dtype=torch.uint8, device=dev)
            self.n_ssim = int(lib.gsx_ssim_partials(Cl, 3, H, W))
            self.loss_rows_fused = int(lib.gsx_ssim_bwd_map_loss_rows(Cl, H, W))
            self.map_ws = torch.empty(max(int(lib.gsx_map_loss_workspace_bytes(Cl, H, W)), self.loss_rows_fused * 24 + 256),
                                      dtype=torch.uint8, device=dev)
            self._one = torch.ones(1, device=dev)
            self._s_r = (C.c_int64 * 4)(H * W * r.CH, 1, W * r.CH, r.CH)
            self._s_g = (C.c_int64 * 4)(H * W * 3, 1, W * 3, 3)
        self.iso_ws = torch.empty(int(lib.gsx_isotropic_workspace_bytes(N)), dtype=torch.uint8, device=dev)
        # ---- the update: six splat tensors + (dt, dR) of every learnable window pose, identical on every rank ------------
        grad_of = {id(getattr(splats, n)): self.grad_views[n] for n in GRAD_PARAMS}
        for i, f in enumerate(self.window):
            if self.learnable[i]:
                grad_of[id(f.pose.dt)] = self.g_dt[i]
                grad_of[id(f.pose.dR)] = self.g_dR[i]
        vis = self.vis_i32 if self.world > 1 else (self.r.vis_count if self.r is not None else self.vis_i32)
        self._vis = vis
        from .optim import AdamPack
        pose_grads = {k: v for k, v in grad_of.items() if k not in {id(getattr(splats, n)) for n in GRAD_PARAMS}}
        # gated on the iteration's overflow flag (device side): a truncated render never reaches the map, the poses, the
        # moments or the step counters - on any rank
        if self.world == 1:
            decay = (splats.opacities, vis, 1, float(conf.opacity_decay)) if decay_opacity else None
            self.adam = AdamPack([optimizers.splat_opt, optimizers.pose_opt], grad_of, decay, gate=self.overflow)
        else:
            # Adam on THIS RANK'S CHUNK of the map: the slices of the six arrays its 1 / G of the flat layout covers, read from
            # the reduce-scattered gradient chunk; the window poses (a few dozen scalars) are updated by every rank alike
            opt = optimizers.splat_opt
            groups = opt.param_groups                      # one group per array, in GRAD_PARAMS order (mapping.SPLAT_LRS)
            if opt._shared_step is None:
                opt._shared_step = torch.full((1,), groups[0]["_host_step"], dtype=torch.int64, device=dev)
            fs = self.flat_state

            def pieces_of(described):
                pieces, decay = [], None
                for k, a, n_el, c_off in described:
                    o = self.bucket.offsets[k] + a
                    pc = dict(p=fs["pflat"][o:o + n_el], g=self.bucket.gchunk[c_off:c_off + n_el], m=fs["mflat"][o:o + n_el],
                              v=fs["vflat"][o:o + n_el], lr=groups[k]["lr"], betas=groups[k]["betas"], eps=groups[k]["eps"],
                              step=opt._shared_step, group=groups[k])
                    pieces.append(pc)
                    if decay_opacity and GRAD_PARAMS[k] == 'opacities':
                        decay = (pc["p"], self.bucket.vis_all[a:a + n_el], 1, float(conf.opacity_decay))
                return pieces, decay

            # ranged exchange: one update launch per range (range 0's also steps the window poses and the step counters), so
            # that the all-gather of a range can follow its update while the next range is still being updated
            self.adam_rest = []
            if self.ranged:
                pieces, decay = pieces_of(self.bucket.pieces_of_range(0))
                self.adam = AdamPack([optimizers.pose_opt], pose_grads, decay, gate=self.overflow, pieces=pieces)
                for k in range(1, self.bucket.ranges):
                    pieces, decay = pieces_of(self.bucket.pieces_of_range(k))
                    self.adam_rest.append(AdamPack([], {}, decay, gate=self.overflow, pieces=pieces, bump=False))
            else:
                pieces, decay = pieces_of(self.bucket.pieces())
                self.adam = AdamPack([optimizers.pose_opt], pose_grads, decay, gate=self.overflow, pieces=pieces)
            for g_ in groups:                              # host bookkeeping follows all six arrays on every rank
                if all(g_ is not h for h in self.adam._groups):
                    self.adam._groups.append(g_)
        self.adam_poses = AdamPack([optimizers.pose_opt], pose_grads) if pose_grads else None
        self.stream = torch.cuda.Stream(device=dev)
        self.graph, self.graph2 = HipGraph(), HipGraph()
        self.steps = 0
        if self.ranged:
            K = self.bucket.ranges
            self.comm = torch.cuda.Stream(device=dev) if self.overlap else None
            self._ev = [torch.cuda.Event() for _ in range(2 * K + 4)]
            self._rows = [(k * self.bucket.Nr, min(N, (k + 1) * self.bucket.Nr)) for k in range(K)]

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def refresh_inputs(self):
        """copies the window's images and exposure parameters into the plan's buffers (they are constants of a window)"""
        for j, i in enumerate(self.mine):
            f = self.window[i]
            self.gt[j].copy_(f.img)
            if f.exposure_params is not None:
                self.exposure[j].copy_(f.exposure_params.detach().reshape(2))

    def matches(self, splats, window) -> bool:
        return (self.r is None or self.r.matches(splats)) and len(window) == self.Cw and \
            all(a is b for a, b in zip(window, self.window)) and int(splats.means.shape[0]) == self.N

    def enqueue_render_backward(self, st: int, keep: bool = False, project: bool = True):
        """render + loss + backward of this rank's cameras: fills the bucket (map gradients, counts source, pose rows).
        project = False (ranged exchange): everything but the projection backward and the PoseZhou backward that reads its
        pose partials - those follow range by range (``_exchange_ranged``)"""
        conf = self.conf
        if self.world > 1:
            # rows of other ranks' cameras and the loss slots hold last iteration's sums: clear them (one small launch)
            tail = self.bucket.tail
            check(lib.gsx_zero_words(_p(tail), tail.numel(), st), "gsx_zero_words")
        r = self.r
        if r is None:
            return
        Cl, H, W = r.C, r.H, r.W
        shard = Cl / float(self.Cw)
        multi = self.world > 1
        w_photo, w_ssim = shard * (1.0 - conf.ssim_weight), shard * conf.ssim_weight
        w_iso = 0.0 if multi else conf.isotropic_regularization_weight    # (multi-rank: added between the collectives)
        w_tv = conf.depth_regularization_weight if self.regularize else 0.0
        mode = 0 if conf.active_gs else 1
        check(lib.gsx_pose_zhou_fwd(Cl, self._Rt, self._dR, self._dt, self._flags, _p(r.viewmats), st),
              "gsx_pose_zhou_fwd")
        r.forward(st)
        n_ssim = 0
        numel_ssim = Cl * 3 * (H - 10) * (W - 10)
        ssim_grad = None
        if w_ssim != 0.0:
            n_ssim = self.n_ssim
            check(lib.gsx_ssim_fwd(_p(r.render), _p(self.gt), Cl, 3, H, W, self._s_r, self._s_g, 5, None,
                                   _p(self.dm[0]), _p(self.dm[1]), _p(self.dm[2]), _p(self.ssim_ws),
                                   self.ssim_ws.numel(), st), "gsx_ssim_fwd")
            if not self.FUSE_SSIM_LOSS:
                check(lib.gsx_ssim_bwd(_p(r.render), _p(self.gt), Cl, 3, H, W, self._s_r, self._s_g, 5, _p(self.dm[0]),
                                       _p(self.dm[1]), _p(self.dm[2]), _p(self._one), -w_ssim / numel_ssim,
                                       _p(self.ssim_grad), st), "gsx_ssim_bwd")
                ssim_grad = self.ssim_grad
        denom = Cl * H * W * (3 if mode == 1 else 1)
        loss_hw = (H, W)
        if n_ssim and self.FUSE_SSIM_LOSS:
            # SSIM backward and the loss block in one pass: the planar SSIM gradient never goes through memory, render and target
            # are read once less, one launch less (csrc/ssim.hip ssim_bwd_loss_kernel); one partial row per 32 x 16 tile
            check(lib.gsx_ssim_bwd_map_loss(_p(r.render), _p(r.alphas), _p(self.gt), _p(self.exposure), Cl, H, W, r.CH,
                                            r.depth_index, r.betas_index, mode, w_photo / denom, w_tv, 0.4, 5, _p(self.dm[0]),
                                            _p(self.dm[1]), _p(self.dm[2]), _p(self._one), -w_ssim / numel_ssim,
                                            _p(r.v_render), _p(self.map_ws), self.map_ws.numel(), st), "gsx_ssim_bwd_map_loss")
            loss_hw = (self.loss_rows_fused // Cl, 256)
        else:
            check(lib.gsx_map_loss(_p(r.render), _p(r.alphas), _p(self.gt), _p(self.exposure), Cl, H, W, r.CH, r.depth_index,
                                   r.betas_index, mode, w_photo / denom, w_tv, 0.4, _p(ssim_grad), None, _p(r.v_render),
                                   None, _p(self.map_ws), self.map_ws.numel(), st), "gsx_map_loss")
        if project:
            r.backward(st, keep)
            self.enqueue_pose_partials(st)
        else:
            r.backward_raster(st)
        iso_ws, n_iso = None, 0
        if w_iso != 0.0:
            iso_ws, n_iso = self.iso_ws, self.N
            check(lib.gsx_isotropic_loss_acc(_p(r.map[2]), _p(r.vis_count), self.N, w_iso, None,
                                             _p(self.grad_views['scales']), _p(iso_ws), iso_ws.numel(), st),
                  "gsx_isotropic_loss_acc")
        pm = 1.0 / denom
        c0 = (C.c_float * 5)(w_photo * pm, w_photo * pm, w_tv, -w_ssim / numel_ssim if n_ssim else 0.0,
                             w_iso if iso_ws is not None else 0.0)
        c1 = (C.c_float * 5)(shard * pm, shard * pm, 0.0, 0.0, 0.0)
        check(lib.gsx_loss_finish(_p(self.map_ws), Cl, loss_hw[0], loss_hw[1], _p(self.ssim_ws) if n_ssim else None, n_ssim,
                                  _p(iso_ws), n_iso, c0, c1, w_ssim if n_ssim else 0.0, 0.0, None, _p(self.g_exposure),
                                  _p(self.out2), st), "gsx_loss_finish")
        # this rank's share of the overflow flag: 1.0 if the render above truncated a tile list (sticky status bit 1), 1024.0 if
        # its tile counts were clamped as corrupt (bit 2): either gates the update on every rank (finish_step tells them apart)
        check(lib.gsx_status_flag(_p(r.status), 1, 3, _p(self.overflow), st), "gsx_status_flag")

    def enqueue_pose_partials(self, st: int):
        """pose partials of the projection backward -> this rank's rows of the window's pose gradients"""
        r = self.r
        if r is not None and any(self.learnable[i] for i in self.mine):
            check(lib.gsx_pose_zhou_bwd_partials(r.C, self._Rt, self._dR, self._dt, self._flags, _p(r.pose_ws),
                                                 r.pose_blocks, None, self._vdR, self._vdt, st),
                  "gsx_pose_zhou_bwd_partials")

    def _isotropic_rows(self, rows, st: int):
        """the isotropic term of rows [g0, g1) added into the local scale gradients (rank 0 only: it enters the sum once)"""
        w = self.conf.isotropic_regularization_weight
        g0, g1 = rows
        if self.rank != 0 or w == 0.0 or g1 <= g0:
            return
        sc, vs, gv = self.splats.scales.detach(), self.vis_i32, self.grad_views['scales']
        check(lib.gsx_isotropic_loss_acc(sc.data_ptr() + 12 * g0, vs.data_ptr() + 4 * g0, g1 - g0, w, None,
                                         gv.data_ptr() + 12 * g0, _p(self.iso_ws), self.iso_ws.numel(), st),
              "gsx_isotropic_loss_acc")

    def _exchange_ranged(self, st: int, keep: bool = False):
        """projection backward range by range on the caller's stream, the gradient exchange of every finished range behind it
        on the second one: [counts all-reduce] | range 0 .. K-1: (isotropic term, staging copy, reduce-scatter) | tail
        all-reduce.  Returns with the caller's stream waiting for all of it."""
        cur = torch.cuda.current_stream(self.dev)
        comm = self.comm if self.overlap else cur
        ev, b = self._ev, self.bucket

        def on_comm(after, fn):
            if comm is cur:
                fn()
                return
            after.record(cur)
            with torch.cuda.stream(comm):
                comm.wait_event(after)
                fn()

        on_comm(ev[0], lambda: b.reduce_counts(None if self.r is None else self.r.vis_count))
        for k, rows in enumerate(self._rows):
            if self.r is not None:
                self.r.backward_project(st, keep, rows=rows)

            def send(k=k, rows=rows):
                self._isotropic_rows(rows, current_stream_ptr(self.dev))
                b.reduce_range(k)
            on_comm(ev[1 + k], send)
        self.enqueue_pose_partials(st)
        on_comm(ev[1 + len(self._rows)], b.reduce_tail)
        if comm is not cur:
            ev[2 + len(self._rows)].record(comm)
            cur.wait_event(ev[2 + len(self._rows)])

    def _update_ranged(self, st: int):
        """Adam range by range on the caller's stream, the all-gather of every updated range behind it on the second one"""
        cur = torch.cuda.current_stream(self.dev)
        comm = self.comm if self.overlap else cur
        K, b = self.bucket.ranges, self.bucket
        ev = self._ev[K + 3:]
        for k, pack in enumerate([self.adam] + self.adam_rest):
            pack.launch(st)
            if comm is cur:
                b.gather_range(self.flat_state["pflat"], k)
                continue
            ev[k].record(cur)
            with torch.cuda.stream(comm):
                comm.wait_event(ev[k])
                b.gather_range(self.flat_state["pflat"], k)
        if comm is not cur:
            ev[K].record(comm)
            cur.wait_event(ev[K])
        self.flat_state["sharded"] = True

    def reduce(self):
        """the gradient exchange of an iteration (eager, on torch's current stream, between the two graphs): all-reduce of the
        head, the isotropic term added ONCE (rank 0) into the local scale gradients from the window-wide visibility, then the
        reduce-scatter of the gradient bucket (gslam_amd.dist.StepBucket.reduce)"""
        if self.world == 1:
            return
        w = self.conf.isotropic_regularization_weight

        def between():
            if self.rank == 0 and w != 0.0:
                check(lib.gsx_isotropic_loss_acc(_p(self.splats.scales.detach()), _p(self.vis_i32), self.N, w, None,
                                                 _p(self.grad_views['scales']), _p(self.iso_ws), self.iso_ws.numel(),
                                                 current_stream_ptr(self.dev)), "gsx_isotropic_loss_acc")
        self.bucket.reduce(None if self.r is None else self.r.vis_count, between)

    def gather_params(self):
        """all-gather of the parameter chunks the ranks' slices of Adam just updated (eager, after the update graph)"""
        if self.world > 1:
            self.bucket.gather(self.flat_state["pflat"], self.flat_state["stage"])
            self.flat_state["sharded"] = True          # from here on a moment is valid on its owner only

    def gather_moments(self):
        """both Adam moments whole on every rank again: before the map is re-packed (pruning, insertion), which moves rows -
        and with them the ownership of a moment - between chunks"""
        if self.world > 1 and self.flat_state["sharded"]:
            self.bucket.gather(self.flat_state["mflat"], self.flat_state["stage"])
            self.bucket.gather(self.flat_state["vflat"], self.flat_state["stage"])
            self.flat_state["sharded"] = False

    def enqueue_update(self, st: int):
        """Adam (+ opacity decay) on the whole map (one rank) or on this rank's chunk of it, and on the window poses"""
        self.adam.launch(st)

    def _enqueue_all(self, st: int):
        self.enqueue_render_backward(st)
        self.enqueue_update(st)

    def prepare(self):
        """capacity probe, ONE eager render + loss + backward (no update: the map and the optimiser state stay as they
        are) and the capture of the step"""
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        st = self.stream.cuda_stream
        if self.r is not None:
            for _ in range(4):
                check(lib.gsx_pose_zhou_fwd(self.r.C, self._Rt, self._dR, self._dt, self._flags, _p(self.r.viewmats), st),
                      "gsx_pose_zhou_fwd")
                self.r.probe(st)
                self.enqueue_render_backward(st)
                self.stream.synchronize()
                if self.r.check_capacity():
                    break
            else:
                raise RuntimeError("tile-list capacity keeps overflowing during warm-up")
        if self.world == 1:
            self.graph.capture(self.stream, self._enqueue_all)
        elif self.ranged:
            self.graph.capture(self.stream, lambda st_: self.enqueue_render_backward(st_, project=False))
        else:
            self.graph.capture(self.stream, self.enqueue_render_backward)
            self.graph2.capture(self.stream, self.enqueue_update)
        if self.r is not None:
            self.r.stale = False
        self.stream.synchronize()
        torch.cuda.current_stream(self.dev).wait_stream(self.stream)

    def step(self, graphed: bool = True):
        """one iteration on torch's current stream; returns (total, photometric) as views of the bucket's loss slots -
        device values, valid once the stream has run (multi-rank: window-wide sums after the reduction).

        CONTRACT: the update is gated ON THE DEVICE by the sticky overflow status of the render: once a tile
        list of this plan has overflowed, every further step() computes gradients and applies NOTHING until the host has
        settled it.  So follow every step() - or every short run of them - with ``finish_step()`` (one read-back of loss and
        flag, grows the lists, tells the caller to redo) or at least ``capacity_ok()``.  The host-side step counters
        (``adam.note_steps``, ``self.steps``) assume the update happened: ``finish_step()`` takes the skipped iteration back
        out of them; ``capacity_ok()`` does NOT - it cannot know how many of the polled steps were gated - so a caller that
        polls must redo its iterations from a state it trusts (the device-side step counters, which the gate also holds,
        stay exact either way: Adam's bias correction reads those)."""
        if graphed and (not self.graph.captured or (self.r is not None and self.r.stale)):
            self.prepare()
        st = current_stream_ptr(self.dev)
        if self.r is not None:
            self.r.clean(st)                                 # (only after a render_backward(): its gradient records)
        if self.world == 1:
            if graphed:
                self.graph.launch(st)
            else:
                if self.r is not None and self.r.capacity == 0:
                    check(lib.gsx_pose_zhou_fwd(self.r.C, self._Rt, self._dR, self._dt, self._flags, _p(self.r.viewmats),
                                                st), "gsx_pose_zhou_fwd")
                    self.r.probe()
                self._enqueue_all(st)
        elif self.ranged:
            if graphed:
                self.graph.launch(st)
            else:
                if self.r is not None and self.r.capacity == 0:
                    check(lib.gsx_pose_zhou_fwd(self.r.C, self._Rt, self._dR, self._dt, self._flags, _p(self.r.viewmats),
                                                st), "gsx_pose_zhou_fwd")
                    self.r.probe()
                self.enqueue_render_backward(st, project=False)
            self._exchange_ranged(st)
            self._update_ranged(st)
        else:
            if graphed:
                self.graph.launch(st)
            else:
                if self.r is not None and self.r.capacity == 0:
                    check(lib.gsx_pose_zhou_fwd(self.r.C, self._Rt, self._dR, self._dt, self._flags, _p(self.r.viewmats),
                                                st), "gsx_pose_zhou_fwd")
                    self.r.probe()
                self.enqueue_render_backward(st)
            self.reduce()
            if graphed:
                self.graph2.launch(st)
            else:
                self.enqueue_update(st)
            self.gather_params()
        self.adam.note_steps(1)
        self.steps += 1
        return self.out2[0], self.out2[1]

    def render_backward(self):
        """render + loss + backward (+ the collective) WITHOUT the update, eagerly on torch's current stream: the
        iteration whose gradients feed densification (backend.py:329-337); ``step_poses()`` then applies what the
        reference's optimiser step still applies on that iteration (the re-packed map tensors have no gradient)"""
        st = current_stream_ptr(self.dev)
        if self.r is not None and self.r.capacity == 0:
            check(lib.gsx_pose_zhou_fwd(self.r.C, self._Rt, self._dR, self._dt, self._flags, _p(self.r.viewmats), st),
                  "gsx_pose_zhou_fwd")
            self.r.probe()
        # the gradient records stay as accumulated (densification reads means2d.grad out of them); the next step() cleans up
        if self.ranged:
            self.enqueue_render_backward(st, keep=True, project=False)
            self._exchange_ranged(st, keep=True)
            return self.out2[0], self.out2[1]
        self.enqueue_render_backward(st, keep=True)
        self.reduce()
        return self.out2[0], self.out2[1]

    def step_poses(self):
        if self.adam_poses is not None:
            self.adam_poses.launch(current_stream_ptr(self.dev))
            self.adam_poses.note_steps(1)

    def decay_opacities(self):
        """backend.py:356-359 as its own launch (for callers that decide about it after reading the loss)"""
        vis = self._vis
        check(lib.gsx_opacity_decay(_p(self.splats.opacities.detach()), _p(vis), self.N, 1,
                                    float(self.conf.opacity_decay), current_stream_ptr(self.dev)), "gsx_opacity_decay")

    def capacity_ok(self) -> bool:
        """LOCAL check of this rank's sticky overflow status (grows the lists and marks the plan stale on overflow).  For
        callers that poll instead of calling finish_step() after every step(): on False, the steps since the overflow
        applied no update (device gate) - how many is not known to the host, the caller redoes its iterations."""
        return True if self.r is None else self.r.check_capacity()

    def finish_step(self):
        """Reads the iteration's (total, photometric, overflow) in ONE blocking read-back - the reference's loss.item() of
        gslam/backend.py:351 - and settles the overflow protocol: the flag is the sum over ranks, so every rank takes the same
        branch.  -> (total, photometric, ok).  ok = False: the iteration applied NO update anywhere (the update launches are
        gated on the flag on the device); the ranks whose lists overflowed have grown them (their next step() re-captures);
        the caller redoes the iteration - on all ranks, or the collectives go out of step."""
        total, pm, flag, _ = self.out4.tolist()
        if flag > 0.0:
            # host bookkeeping of the skipped update (step() counted it)
            self.adam.note_steps(-1)
            self.steps -= 1
            self.capacity_ok()          # grows this rank's lists on overflow; RAISES on the rank whose tile counts are corrupt
            if flag >= 1024.0:
                # some OTHER rank built its lists from corrupt counts (it has raised above): no rank may go on - a redo would
                # leave the collectives of the surviving ranks waiting for it
                raise RuntimeError("corrupt tile counts in a render plan of another rank (overflow flag "
                                   f"{flag:g}): the iteration applied no update; aborting on every rank")
            return total, pm, False
        return total, pm, True

    def as_output(self):
        """RasterizationOutput over this rank's cameras (None for a rank without cameras)"""
        if self.r is None:
            return None
        out = self.r.as_output()
        out._window_cams = self.Cw
        out._window_index = list(self.mine)          # row j of the per-camera arrays = window camera mine[j]
        return out

    def depthmaps(self):
        """[C_local,H,W] accumulated depth of the last render (what backend.py:361-362 stores as est_depths)"""
        return None if self.r is None else self.r.render[..., self.r.depth_index]
