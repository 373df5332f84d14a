"""Frontend <-> backend message tags, value-compatible with gslam/messages.py:4-12 (StrEnum + auto() = lower-case
member names).  Tuple shapes kept from the reference (SURVEY.md §8b vii):
  F->B: (REQUEST_INIT, Frame) | (ADD_FRAME, Frame) | None
  B->F: (SYNC, keyframes, depthmap[H,W], rgbs[H,W,3], splats(no-grad clone), pose_graph) | (END_SYNC, splats, keyframes)
"""
from enum import Enum


class _StrEnum(str, Enum):
    def __str__(self) -> str:
        return str(self.value)


class FrontendMessage(_StrEnum):
    ADD_FRAME = "add_frame"
    ADD_REFINED_DEPTHMAP = "add_refined_depthmap"
    REQUEST_INIT = "request_init"


class BackendMessage(_StrEnum):
    SYNC = "sync"
    END_SYNC = "end_sync"
