"""The window pose refiner's closure chain (26 closures over the 8-camera window against the 500 k map) on an idle GPU:
time per closure; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gslam_amd.map import GaussianSplattingData  # noqa: E402
from gslam_amd.mapping import GraphedPoseRefiner  # noqa: E402
from gslam_amd.synthetic import make_scene  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    N, W, H = int(os.environ.get("N", 500_000)), 640, 480
    gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
    frames, cam = bench.make_frames(list(range(bench.WINDOW)), W, H, dev, gt)
    del gt
    splats = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    for i, f in enumerate(frames):
        f.index = i
        f.exposure_params = f.exposure_params.detach()
    if os.environ.get("FUSED"):                         # the closure on the tracking machinery instead of the generic chain - A/B
        import gslam_amd.plan as P1
        o1 = P1.WindowClosure.__init__

        def p1(self, *a, **k):
            k["fused"] = True
            o1(self, *a, **k)
        P1.WindowClosure.__init__ = p1
    if os.environ.get("FRONT"):                         # force the fused front (default: generic path when C * N >= 2^20)
        import gslam_amd.plan as P
        orig = P.RenderPlan.__init__

        def patched(self, *a, **k):
            k["front"] = bool(int(os.environ["FRONT"]))
            orig(self, *a, **k)
        P.RenderPlan.__init__ = patched
    ref = GraphedPoseRefiner(splats, frames)
    ref.capture()
    for _ in range(2):
        ref.run_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ref.run_async()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"refinement {el / reps * 1e3:.2f} ms = {ref.max_eval + 1} closures x {el / reps / (ref.max_eval + 1) * 1e6:.1f} us; "
          f"fused={getattr(ref.plan, 'fused', False)} front={ref.plan.r.front} compact={ref.plan.r.compact} "
          f"M={ref.plan.r.last_M} ok={ref.capacity_ok()}")
    del ref, splats, frames        # graphs are destroyed before the interpreter (and a profiler) shuts down


if __name__ == "__main__":
    main()
