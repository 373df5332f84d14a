"""The tracking frontend behind the reference's message API (gslam/frontend.py:64-520) without its shell (rerun
logging, evaluation, sensors, checkpoints): frame intake, constant-motion pose prediction, the Gaussian-splatting
tracker, and the SYNC / END_SYNC handling.

  out: (REQUEST_INIT, Frame) | (ADD_FRAME, Frame) | None          in: (SYNC, ...) | (END_SYNC, ...)

Tracking is ``igs_track_lbfgs`` (frontend.py:604-662) run by ``tracking.GraphedTracker``: the closure is one HIP
graph, the Adam + L-BFGS logic a device state machine, so a frame costs no read-back.  The tracker is re-captured
whenever a SYNC replaces the frontend's copy of the map (new tensors, possibly a new N)."""
from __future__ import annotations

import time
from copy import deepcopy
from typing import Dict, List, Optional

import torch

from .map import GaussianSplattingData
from .messages import BackendMessage, FrontendMessage
from .primitives import Frame, PoseZhou
from .tracking import GraphedTracker, TrackingConfig


class Frontend:
    def __init__(self, conf: TrackingConfig, backend_queue, frontend_queue, sensor_queue, frontend_done_event=None,
                 backend_done_event=None, global_pause_event=None):
        self.conf = conf
        self.map_queue = backend_queue
        self.queue = frontend_queue
        self.sensor_queue = sensor_queue
        self.frontend_done_event = frontend_done_event
        self.backend_done_event = backend_done_event
        self.global_pause_event = global_pause_event
        self.keyframes: Dict[int, Frame] = dict()
        self.frames: List[Frame] = []
        self.pose_graph = None
        self.splats: Optional[GaussianSplattingData] = None
        self.tracker: Optional[GraphedTracker] = None
        self.reference_frame = self.reference_depthmap = self.reference_rgbs = None
        self.waiting_for_sync = self.waiting_for_end_sync = self.done = False
        self.last_losses: List[float] = []

    # ---- frontend.py:149-171 -----------------------------------------------------------------------------------------
    def initialize(self, new_frame: Frame):
        new_frame.pose = PoseZhou(torch.eye(4, device=self.conf.device)).to(self.conf.device)
        self.keyframes[new_frame.index] = new_frame
        self.reference_frame = new_frame
        self.reference_rgbs = new_frame.img
        new_frame.exposure_params = torch.zeros([2], device=new_frame.img.device)
        self.request_initialization(new_frame)

    def request_initialization(self, f: Frame):
        self.map_queue.put((FrontendMessage.REQUEST_INIT, deepcopy(f)))
        self.waiting_for_sync = True

    def add_frame_to_backend(self, new_frame: Frame):
        self.map_queue.put((FrontendMessage.ADD_FRAME, deepcopy(new_frame)))

    # ---- frontend.py:173-250 -----------------------------------------------------------------------------------------
    def track(self, new_frame: Frame):
        if len(self.frames) == 0:
            self.initialize(new_frame)
            self.frames.append(new_frame.strip())
            return new_frame.pose()
        if len(self.frames) == 1:
            pose = self.frames[-1].pose()
        else:                                                   # constant motion model
            pose_a, pose_b = self.frames[-2].pose(), self.frames[-1].pose()
            pose = pose_b @ torch.linalg.inv(pose_a) @ pose_b
        new_frame.exposure_params = torch.zeros([2], device=new_frame.img.device)
        new_frame.pose = PoseZhou(pose.detach()).to(self.conf.device)
        if self.tracker is None:
            self.tracker = GraphedTracker(self.splats, new_frame.camera, self.conf)
        prev_exposure = self.frames[-1].exposure_params if self.conf.learn_exposure_params else None
        loss, _n = self.tracker.track(new_frame, prev_exposure)
        self.last_losses.append(loss)
        self.frames.append(new_frame.strip())
        if new_frame.index > 0:
            self.add_frame_to_backend(new_frame)
        return new_frame.pose()

    # ---- frontend.py:253-273 -----------------------------------------------------------------------------------------
    def sync(self, keyframes, depthmap, rgbs, splats: GaussianSplattingData, pose_graph):
        self.keyframes = deepcopy(keyframes)
        self.reference_depthmap = depthmap.clone()
        self.reference_frame = self.keyframes[sorted(self.keyframes.keys())[-1]]
        self.reference_rgbs = rgbs
        if splats.means.is_cuda:
            # one-launch copy into the frontend's own map; same N -> same tensors, the captured tracking graph survives
            from .transport import receive
            self.splats, replaced = receive(getattr(self, "splats", None), splats)
        else:
            self.splats, replaced = deepcopy(splats), True
        self.pose_graph = pose_graph
        if replaced:
            self.tracker = None                                 # new map tensors: the captured closure is stale

    def sync_at_end(self, splats: GaussianSplattingData, keyframes):
        self.splats, self.keyframes = splats, deepcopy(keyframes)

    def handle_message_from_backend(self, message):
        tag = message[0]
        if tag == BackendMessage.SYNC:
            _, keyframes, depthmap, rgbs, splats, pose_graph = message
            self.sync(keyframes, depthmap, rgbs, splats, pose_graph)
            self.waiting_for_sync = False
        elif tag == BackendMessage.END_SYNC:
            _, map_data, keyframes = message
            self.sync_at_end(map_data, keyframes)
            self.waiting_for_end_sync = False
            self.done = True
        else:
            raise ValueError(f"Unknown message_from_map={message!r}")

    # ---- frontend.py:432-520 -----------------------------------------------------------------------------------------
    def run(self, timeout_s: float = 3000.0):
        last_heard = time.time()
        while True:
            if not self.queue.empty():
                self.handle_message_from_backend(self.queue.get())
                last_heard = time.time()
            if self.waiting_for_end_sync:
                if (time.time() - last_heard) > timeout_s:
                    break
                time.sleep(0.001)
                continue
            if self.waiting_for_sync:
                time.sleep(0.001)
                continue
            if self.done:
                break
            if self.sensor_queue.empty():
                time.sleep(0.001)
                continue
            frame = self.sensor_queue.get()
            if frame is None:                                   # data stream exhausted
                self.map_queue.put(None)
                self.waiting_for_end_sync = True
                last_heard = time.time()
                continue
            self.track(frame.to(self.conf.device))
        if self.backend_done_event is not None:
            self.backend_done_event.wait(timeout=timeout_s)
        if self.frontend_done_event is not None:
            self.frontend_done_event.set()
