"""The mapping backend behind the reference's frontend <-> backend message API (gslam/backend.py:110-899), without
its shell (rerun / viser logging, point-cloud dumps, CLI): keyframe selection, map initialisation and growth, the
bundle-adjustment loop, pruning, window pose refinement and the SYNC / END_SYNC payloads.

Messages are the reference's tuples (gslam_amd/messages.py):
  in : (REQUEST_INIT, Frame) | (ADD_FRAME, Frame) | None
  out: (SYNC, keyframes, depthmap [H,W], rgbs [H,W,3], splats (no-grad clone), pose_graph) | (END_SYNC, splats, keyframes)

``Backend`` is a plain object with the reference's method names; ``run()`` is the reference's loop and works on any
pair of queues with ``get / put / empty`` (``queue.Queue`` between threads of one process - one process per GPU, the
frontend on another HIP stream - or ``torch.multiprocessing.Queue`` between processes as in main.py:61-95).
Everything heavy is the HIP path: renders through gslam_amd.rasterization, BA steps through mapping.BundleAdjuster,
map surgery through the one-launch kernels of insertion.py / pruning.py."""
from __future__ import annotations

import math
import random
import time
from collections import defaultdict
from copy import deepcopy
from dataclasses import dataclass
from itertools import combinations
from typing import Dict, List, Optional

import torch

from .insertion import InsertFromDepthMap, InsertUsingImagePlaneGradients
from .map import GaussianSplattingData
from .mapping import BundleAdjuster, MapConfig as _CoreConfig, optimize_poses_lbfgs
from .messages import BackendMessage, FrontendMessage
from .primitives import Frame, PoseZhou
from .pruning import PruneIllConditionedGaussians, PruneLargeGaussians, PruneLowOpacity, prune_using_mask
from .rasterization import RasterizationOutput
from .utils import StopOnPlateau


def add_constraint(pose_graph, kf1: int, kf2: int):
    """gslam/pose_graph.py:6-9"""
    pose_graph[kf1].add(kf2)
    pose_graph[kf2].add(kf1)
    return pose_graph


def remove_keyframe(pose_graph, kf_id: int):
    """gslam/pose_graph.py:12-16"""
    del pose_graph[kf_id]
    for kf in pose_graph:
        pose_graph[kf].discard(kf_id)
    return pose_graph


@dataclass
class MapConfig(_CoreConfig):
    """gslam/backend.py:43-107 (the fields the loop reads; the loss / optimiser ones are inherited)."""
    initial_opacity: float = 0.3
    initial_scale: float = 1.0
    optim_window_random_keyframes: int = 2
    opacity_pruning_threshold: float = 0.2
    size_pruning_threshold: int = 256
    enable_pgo: bool = False
    kf_cov: float = 0.9
    kf_oc: float = 0.99
    kf_m: float = 0.15
    kf_cos: float = math.cos(math.pi / 30)
    use_gt_depths: bool = False
    seed: int = 0                       # multi-rank only: common seed of the replicas' random draws
    device_pose_refiner: bool = True    # window pose L-BFGS on the device (False: torch.optim.LBFGS on the host)
    densify_every: int = 200            # backend.py:329 (`total_step % 200`)
    sync_every: int = 5                 # backend.py:864 (`frame.index % 5`)


class Backend:
    def __init__(self, conf: MapConfig, queue, frontend_queue, backend_done_event=None, global_pause_event=None):
        self.conf = conf
        self.queue = queue
        self.frontend_queue = frontend_queue
        self.backend_done_event = backend_done_event
        self.keyframes: Dict[int, Frame] = dict()
        self.frames: List[Frame] = []
        self.splats = GaussianSplattingData.empty(conf.device)
        self.pruning_opacity = PruneLowOpacity(conf.opacity_pruning_threshold)
        self.pruning_size = PruneLargeGaussians(conf.size_pruning_threshold)
        self.insertion_depth_map = InsertFromDepthMap(0.1 * conf.initial_scale, 0.2 * conf.initial_scale, 0.1,
                                                      conf.initial_opacity, False, global_pause_event=global_pause_event)
        self.insertion_3dgs = InsertUsingImagePlaneGradients(0.0002, 0.01)
        self.pruning_conditioning = PruneIllConditionedGaussians(3)
        self.pose_graph = defaultdict(set)
        self.total_step = 0
        self.pause_map_optim = False
        self.ba: Optional[BundleAdjuster] = None
        # keyframe-sharded mapping (one Backend replica per rank): every replica makes the same random draws
        # (window sampling, depth-map insertion, densification splits), so their maps stay identical between collectives
        import torch.distributed as td
        if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
            random.seed(int(getattr(conf, "seed", 0)))
            torch.manual_seed(int(getattr(conf, "seed", 0)))
        self.last_outputs: Optional[RasterizationOutput] = None
        self.last_kf_depthmap = self.last_kf_rgbs = None

    # ---- optimisers: the six splat Adams + the pose Adam of backend.py:565-602 are one multi-tensor FusedAdam -------
    @property
    def splat_optimizers(self):
        return None if self.ba is None else self.ba.optimizers

    def initialize_optimizers(self):
        self.ba = BundleAdjuster(self.splats, self.conf, capturable=True)

    # ---- window selection (backend.py:193-247) -----------------------------------------------------------------------
    def optimization_window(self) -> List[Frame]:
        conf = self.conf
        window_size_total = conf.optim_window_last_n_keyframes + conf.optim_window_random_keyframes
        if conf.enable_pgo:
            latest = sorted(self.keyframes.keys())[-1]
            window = {latest}
            neighbors = self.pose_graph[latest]
            if 0 < len(neighbors) < window_size_total:
                window.update(random.sample(sorted(neighbors), min(len(neighbors), window_size_total)))
            elif 0 < len(neighbors):
                window.update(list(neighbors))
            for _ in range(window_size_total - len(window)):
                if len(neighbors) == 0:
                    break
                nn = self.pose_graph[random.sample(sorted(neighbors), 1)[0]]
                if len(nn) == 0:
                    continue
                pick = random.sample(sorted(nn), 1)[0]
                if pick not in window:
                    window.add(pick)
        else:
            n_last = min(len(self.keyframes), conf.optim_window_last_n_keyframes)
            window = list(self.keyframes.keys())[-n_last:]          # (the reference's random part is min(0, .) = 0)
        return [self.keyframes[i] for i in sorted(window)]

    # ---- mapping (backend.py:249-407) --------------------------------------------------------------------------------
    def optimize_map(self, n_iters: Optional[int] = None, prune: bool = True, regularize: bool = True):
        """backend.py:249-407.  An iteration is one replay of the window's launch plan (gslam_amd.plan.MappingStep: render,
        loss, backward, [all-reduce], six splat Adams + pose Adam in one launch); the photometric term is read back once
        per iteration for the early stop, as in the reference (:351), and the opacity decay (:356-359) follows that
        decision as its own launch.  Multi-GPU: every rank runs this loop on its replica; the plan shards the window's
        cameras and its one all-reduce makes update, loss and early-stop decision identical on all ranks."""
        conf = self.conf
        n_iters = conf.num_iters_mapping if n_iters is None else n_iters
        early_stopper = StopOnPlateau(3, 0.012)
        window = self.optimization_window()
        plan = None
        for _ in range(n_iters):
            self.total_step += 1
            window = self.optimization_window()
            plan = self.ba.plan(window, regularize, decay_opacity=False)
            if (self.total_step % conf.densify_every) == 0:
                # densification by image-plane gradients (:329-337): render + loss + backward, grow the map from
                # means2d.grad, then the reference's optimiser step - the re-packed map tensors carry no gradient, so
                # only the poses move
                plan.render_backward()
                self._densify(plan)
                plan.step_poses()
                pm = float(plan.out2[1].item())
                self.ba.map_changed()
                prune = False
                decay = False          # the reference's mask has the old N here (it would fail on a grown map)
            else:
                for _attempt in range(4):
                    plan.step()
                    # backend.py:351 (the sync the reference has too): loss + the iteration's overflow flag in one read.
                    # The flag is summed over ranks by the iteration's all-reduce and gates the update launches on the
                    # device: an iteration whose tile lists were truncated on ANY rank changed nothing on EVERY rank
                    # (map, poses, moments, step counters), and every rank redoes it - the collectives stay in step.
                    _total, pm, ok = plan.finish_step()
                    if ok:
                        break
                else:
                    raise RuntimeError("tile lists kept overflowing in optimize_map")
                decay = True
            if early_stopper.stop(pm):
                self.pause_map_optim = True
                break
            if decay:
                plan.decay_opacities()
        outputs = self._window_outputs(plan, window)
        sharded = self.ba.shard.world_size > 1 and plan is not None and plan.matches(self.splats, window)
        if not sharded:
            if outputs is not None:
                for f, d in zip(window, outputs.depthmaps):
                    f.est_depths = d.detach().clone()
        else:
            # every replica needs every keyframe's depth map (insertion.py:245-277 tests new splats against them): the rank
            # that rendered a camera broadcasts its row (1.2 MB each at 640x480, once per optimize_map call)
            import torch.distributed as td
            local = {i: d for i, d in zip(plan.mine, outputs.depthmaps)} if outputs is not None else {}
            for i, f in enumerate(window):
                buf = local[i].detach().clone() if i in local else torch.empty(plan.H, plan.W, device=self.splats.means.device)
                td.broadcast(buf, src=i % self.ba.shard.world_size, group=self.ba.shard.group)
                f.est_depths = buf
        if prune:
            self._prune(outputs, len(window) >= 2)
        self._render_last_keyframe()

    def _window_outputs(self, plan, window):
        """the last render of the window: the plan's buffers while they still describe the map, a fresh render otherwise
        (the map was re-packed by the densification of the last iteration)"""
        if plan is not None and plan.matches(self.splats, window):
            return plan.as_output()
        with torch.no_grad():
            return self.splats([f.camera for f in window], [f.pose for f in window], render_depth=True,
                               need_n_touched=self.ba.need_n_touched)

    def _densify(self, plan):
        outputs = plan.as_output()
        shard = self.ba.shard
        if outputs is None:                                         # a rank without cameras: zeros into the reduction
            n = self.splats.means.shape[0]
            dev = self.splats.means.device
            outputs = RasterizationOutput(radii=torch.zeros(0, n, dtype=torch.int32, device=dev),
                                          means2d=torch.zeros(0, n, 2, device=dev), width=plan.W, height=plan.H,
                                          n_cameras=0)
            outputs.means2d.grad = torch.zeros(0, n, 2, device=dev)
        self.ba.sync_moments()
        self.insertion_3dgs.step(self.splats, self.splat_optimizers, outputs, None, None,
                                 window_cameras=plan.Cw, reduce_sum=shard.all_reduce_sum if shard.world_size > 1 else None)

    def _prune(self, outputs: Optional[RasterizationOutput], visibility_ok: bool):
        """size / opacity (/ conditioning) pruning, backend.py:364-392 and :409-437.  Multi-GPU: the per-Gaussian
        statistics over the window's cameras are reduced over ranks first (max of the screen radii, sum of the
        ill-conditioned view counts), so that every replica removes the same Gaussians."""
        conf = self.conf
        shard = self.ba.shard
        n = self.splats.means.shape[0]
        dev = self.splats.means.device
        if outputs is not None and outputs.radii.shape[0] > 0:
            radii = outputs.radii[:, :n]
            max_radii = torch.max(radii, dim=0).values
        else:
            radii = torch.zeros(0, n, dtype=torch.int32, device=dev)
            max_radii = torch.zeros(n, dtype=torch.int32, device=dev)
        shard.all_reduce_max(max_radii)
        remove = torch.zeros(n, dtype=torch.bool, device=dev)
        if conf.enable_visibility_pruning and visibility_ok:
            k = conf.optim_window_last_n_keyframes
            nt = None if outputs is None else outputs.n_touched
            if nt is not None and radii.shape[0] > 0:
                # the reference's statistic covers the first k keyframes of the WINDOW (backend.py:370-375); a rank holds the
                # window cameras r, r + G, ...: select its rows by their window index, so that a G-rank run sums over the same
                # camera set as a single-rank run
                widx = getattr(outputs, "_window_index", None)
                rows = list(range(min(k, radii.shape[0]))) if widx is None else [j for j, i in enumerate(widx) if i < k]
                rows_t = torch.tensor(rows, dtype=torch.long, device=dev)
                bad_views = ((radii[rows_t] > 0) & (nt[rows_t][:, :n] == 0)).sum(dim=0).to(torch.int32)
                replicated = widx is None          # run_pruning(): every rank rendered the SAME camera (no window shard)
            else:
                bad_views = torch.zeros(n, dtype=torch.int32, device=dev)
                replicated = False
            if not replicated:                     # a replicated render is already the whole camera set: summing it over the
                shard.all_reduce_sum(bad_views)    # ranks would count every bad view G times (threshold G times tighter)
            remove |= bad_views > self.pruning_conditioning.max_frames_thing      # = PruneIllConditionedGaussians.step
        self.ba.sync_moments()                      # (multi-GPU: whole moments on every rank before the rows move)
        remove |= self.pruning_size.step(self.splats, self.splat_optimizers, max_radii)
        remove |= self.pruning_opacity.step(self.splats, self.splat_optimizers)
        if prune_using_mask(self.splats, self.splat_optimizers, ~remove) > 0:
            self.ba.map_changed()

    def _render_last_keyframe(self):
        last_kf = list(self.keyframes.values())[-1]
        with torch.no_grad():
            outputs = self.splats([last_kf.camera], [last_kf.pose], True)
        last_kf.visible_gaussians = outputs.radii.sum(dim=0) > 0
        self.last_outputs = outputs
        self.last_kf_depthmap = outputs.depthmaps[0]
        self.last_kf_rgbs = outputs.rgbs[0]

    def run_pruning(self):
        last_kf = list(self.keyframes.values())[-1]
        with torch.no_grad():
            outputs = self.splats([last_kf.camera], [last_kf.pose], True)
        self._prune(outputs, len(self.keyframes) >= 2)
        self._render_last_keyframe()

    def optimize_poses_lbfgs(self):
        """backend.py:447-506.  On the GPU the L-BFGS state machine runs on the device (mapping.GraphedPoseRefiner, one
        captured closure per window composition); the host version remains for CPU tensors and oversized windows."""
        window = self.optimization_window()
        loss = self._refine_window_poses(window)
        self._hand_back_poses(window)
        return loss

    def _hand_back_poses(self, window):
        """Multi-GPU: the refinement above runs replicated (every rank renders the whole window), and its backward pass sums
        with floating-point atomics whose order differs from run to run - the replicas' poses could drift apart in the last
        bits.  Rank 0's result is handed to all of them: one broadcast of 25 floats per window pose."""
        shard = None if self.ba is None else self.ba.shard
        if shard is None or shard.world_size == 1 or not window:
            return
        with torch.no_grad():
            buf = torch.stack([torch.cat([f.pose.Rt.reshape(-1), f.pose.dR.reshape(-1), f.pose.dt.reshape(-1)])
                               for f in window]).contiguous()
            shard.broadcast_([buf], src=0)
            for f, row in zip(window, buf):
                f.pose.Rt.copy_(row[:16].view_as(f.pose.Rt))
                f.pose.dR.copy_(row[16:22].view_as(f.pose.dR))
                f.pose.dt.copy_(row[22:25].view_as(f.pose.dt))

    def _refine_window_poses(self, window):
        learn = [x for x in window if x.index != 0]
        if (not self.splats.means.is_cuda or not learn or 9 * len(learn) > 80
                or not getattr(self.conf, "device_pose_refiner", True)):
            return optimize_poses_lbfgs(self.splats, window, self.conf)
        # one refiner (buffers + captured closure) per window SHAPE over the current map tensors: the window's poses,
        # images and exposure are copied into its slots, so sliding the window does not re-capture; a re-packed map does
        refiner = getattr(self, "_pose_refiner", None)
        if refiner is None or not refiner.matches(self.splats, window):
            from .mapping import GraphedPoseRefiner
            refiner = GraphedPoseRefiner(self.splats, window, self.conf)
            self._pose_refiner = refiner
        return refiner.run(window)[0]

    # ---- messages out (backend.py:508-552) ---------------------------------------------------------------------------
    def sync(self):
        """backend.py:508-519.  The map travels as a view of a double-buffered device slot filled by one launch
        (gslam_amd.transport.MapMailbox) instead of seven clones; the tuple keeps the reference's shape."""
        import queue as _queue
        if self.splats.means.is_cuda and isinstance(self.frontend_queue, _queue.Queue):
            # same process (one process per GPU, the frontend on another thread and stream): the double-buffered mailbox
            if getattr(self, "_mailbox", None) is None:
                from .transport import MapMailbox
                self._mailbox = MapMailbox()
            payload = self._mailbox.publish(self.splats)
        else:
            # another process on the other end (the reference's topology, main.py:61-91: torch.multiprocessing queues): the
            # reference's own payload, a no-grad clone; it crosses the process boundary as device memory (HIP IPC handles,
            # torch.multiprocessing's reductions order the consumer behind the producer's stream) - no host copy, and no
            # slot that could be re-used under a lagging consumer
            payload = self.splats.no_grad_clone()
        self.frontend_queue.put((BackendMessage.SYNC, deepcopy(self.keyframes), self.last_kf_depthmap.detach(),
                                 self.last_kf_rgbs.detach(), payload, deepcopy(self.pose_graph)))

    def end_sync(self):
        self.frontend_queue.put((BackendMessage.END_SYNC, self.splats.clone(), deepcopy(self.keyframes)))

    # ---- map life cycle (backend.py:604-670) -------------------------------------------------------------------------
    def initialize(self, frame: Frame):
        conf = self.conf
        frame = frame.to(conf.device)
        self.frames.append(frame.strip())
        self.keyframes[frame.index] = frame
        self.splats = GaussianSplattingData.empty(conf.device).to(conf.device)
        self.initialize_optimizers()
        self.pose_graph[frame.index] = set()
        H, W, _ = frame.img.shape
        mock_depth = torch.ones((1, H, W), device=conf.device)
        mock_depth = (mock_depth + (torch.randn_like(mock_depth) - 0.5) * 0.3) * conf.initial_scale
        mock_alphas = torch.ones((1, H, W, 1), device=conf.device) * 0.01
        mock_outputs = RasterizationOutput(None, mock_alphas, mock_depth)
        self.insertion_depth_map.step(self.splats, self.splat_optimizers, mock_outputs, frame, 5000,
                                      keyframes=list(self.keyframes.values()),
                                      gt_depthmap=frame.gt_depth if conf.use_gt_depths else None)
        self.ba.map_changed()
        # the reference never hands the first keyframe's pose to its pose optimiser (only add_keyframe does,
        # backend.py:665-670): it stays fixed, here by not being trainable
        for p in frame.pose.parameters():
            p.requires_grad_(False)

    def add_keyframe(self, frame: Frame):
        conf = self.conf
        with torch.no_grad():
            outputs = self.splats([frame.camera], [frame.pose], render_depth=True)
        outputs.depthmaps = outputs.depthmaps * conf.initial_scale
        self.ba.sync_moments()
        self.insertion_depth_map.step(self.splats, self.splat_optimizers, outputs, frame, N=100,
                                      keyframes=list(self.keyframes.values()))
        self.ba.map_changed()
        new_frame = Frame(img=frame.img.clone(), timestamp=frame.timestamp, camera=frame.camera.clone(),
                          pose=PoseZhou(frame.pose().detach()).to(conf.device), gt_pose=frame.gt_pose,
                          gt_depth=frame.gt_depth, img_file=frame.img_file, index=frame.index, est_depths=outputs.depths,
                          exposure_params=frame.exposure_params.detach().clone().requires_grad_(False))
        self.keyframes[new_frame.index] = new_frame
        self.ba.optimizers.add_pose(new_frame.pose)
        if len(self.keyframes) >= 1:
            add_constraint(self.pose_graph, *(list(self.keyframes.keys())[-2:]))

    # ---- keyframe / loop-closure tests (backend.py:672-786) ----------------------------------------------------------
    def to_add_pg_edge(self, previous_keyframe: Frame, new_frame: Frame):
        inter = torch.logical_and(new_frame.visible_gaussians, previous_keyframe.visible_gaussians)
        union = torch.logical_or(new_frame.visible_gaussians, previous_keyframe.visible_gaussians)
        return (inter.sum() / union.sum()).item() > self.conf.kf_cov

    def to_remove_keyframe(self, kf_i: Frame, kf_j: Frame):
        inter = torch.logical_and(kf_j.visible_gaussians, kf_i.visible_gaussians)
        oc = inter.sum() / min(kf_i.visible_gaussians.sum().item(), kf_j.visible_gaussians.sum().item())
        return oc.item() > self.conf.kf_oc, oc.item()

    @torch.no_grad()
    def add_pgo_constraints(self):
        for kf in self.keyframes.values():
            kf.visible_gaussians = self.splats([kf.camera], [kf.pose]).radii.sum(dim=0) > 0
        for i, j in combinations(sorted(self.keyframes), 2):
            if i not in self.keyframes or j not in self.keyframes or j in self.pose_graph[i]:
                continue
            if self.to_add_pg_edge(self.keyframes[i], self.keyframes[j]):
                add_constraint(self.pose_graph, i, j)
        for kf in self.keyframes.values():
            kf.visible_gaussians = None

    @torch.no_grad()
    def to_insert_keyframe(self, previous_keyframe: Frame, new_frame: Frame):
        outputs = self.splats([new_frame.camera, previous_keyframe.camera], [new_frame.pose, previous_keyframe.pose],
                              render_depth=True)
        pose_difference = torch.linalg.inv(new_frame.pose()) @ previous_keyframe.pose()
        translation = pose_difference[:3, 3].pow(2.0).sum().pow(0.5).item()
        seen = outputs.alphas[..., 0] > 0.1
        median_depth = outputs.depthmaps[seen].median() if bool(seen.any()) else outputs.depthmaps.median()
        if translation > self.conf.kf_m * median_depth:
            return True
        cosine_sim = torch.nn.functional.cosine_similarity(new_frame.pose()[:3, 2], previous_keyframe.pose()[:3, 2],
                                                           dim=0)
        return bool(cosine_sim < self.conf.kf_cos)

    # ---- the loop (backend.py:818-899) -------------------------------------------------------------------------------
    def handle(self, message) -> bool:
        """one message of the reference's `match`; returns False on the terminating None"""
        if message is None:
            return False
        tag = message[0]
        if tag == FrontendMessage.ADD_REFINED_DEPTHMAP:
            raise NotImplementedError()
        if tag == FrontendMessage.ADD_FRAME:
            frame = deepcopy(message[1])
            self.frames.append(frame.strip())
            if len(self.keyframes) == 0:
                self.initialize(frame)
                return True
            last_keyframe = self.keyframes[sorted(self.keyframes.keys())[-1]]
            if self.to_insert_keyframe(last_keyframe, frame):
                self.pause_map_optim = False
                self.add_keyframe(frame)
                self.optimize_map(1, prune=True, regularize=False)
                if self.conf.enable_pgo:
                    self.add_pgo_constraints()
            if frame.index % self.conf.sync_every == 0:
                self.sync()
            return True
        if tag == FrontendMessage.REQUEST_INIT:
            frame = deepcopy(message[1])
            self.frames.append(frame.strip())
            self.pause_map_optim = False
            self.initialize(frame)
            self.optimize_map(self.conf.num_iters_initialization, False, True)
            self.sync()
            return True
        return True                                              # unknown message: ignored like the reference (logged there)

    def idle_step(self):
        """what the reference does while its queue is empty (backend.py:836-844)"""
        if self.pause_map_optim or len(self.keyframes) == 0:
            return False
        self.optimize_map()
        if len(self.keyframes) > 1:
            self.run_pruning()
            self.optimize_poses_lbfgs()
        return True

    def run(self):
        from ._sync import capture_lock
        self.pause_map_optim = False
        while True:
            if self.queue.empty():
                with capture_lock:
                    busy = self.idle_step()
                if not busy:
                    time.sleep(0.03)
                    continue
            message = self.queue.get()
            with capture_lock:
                if not self.handle(message):
                    break
        self.end_sync()
        if self.backend_done_event is not None:
            self.backend_done_event.set()
