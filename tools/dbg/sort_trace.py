"""Phase stamps of tile_sort_deep_kernel (DIAGNOSTIC build: isect_bin.hip with -DGSX_WG_TRACE) on BASELINE.json configs[4]:
thread 0 of the first 4096 tile workgroups stamps s_memrealtime (100 MHz) at the phase boundaries; printed as the mean
duration of each phase in microseconds."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd.csrc import build  # noqa: E402

build.SOURCES["isect_bin.hip"] = build.SOURCES["isect_bin.hip"] + ["-DGSX_WG_TRACE"] + os.environ.get("EXTRA_DEFS", "").split()
os.remove(os.path.join(build.OBJ, "isect_bin.o"))
build.build()
from gslam_amd import _lib  # noqa: E402
from gslam_amd.rendering import rasterization as gs_rasterization  # noqa: E402
from gslam_amd.synthetic import make_cameras, make_scene  # noqa: E402

dev = torch.device("cuda:0")
N, Cn, W, H = (5_000_000, 1, 1920, 1080) if not os.environ.get("BA2M") else (2_000_000, 8, 640, 480)
sc = make_scene(N, 0)
viewmats, Ks = make_cameras(Cn, W, H)
p = {k: v.to(dev) for k, v in sc.items()}
buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
setter = _lib.lib.gsx_debug_sort_trace
setter.argtypes, setter.restype = [C.c_void_p], C.c_int


def render():
    with torch.no_grad():
        return gs_rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                torch.sigmoid(p["colors"]), viewmats.to(dev), Ks.to(dev), W, H, packed=False)


render()
torch.cuda.synchronize()
assert setter(buf.data_ptr()) == 0
render()
torch.cuda.synchronize()
assert setter(None) == 0
a = buf.cpu().numpy().reshape(4096, 8).astype(np.float64)
a = a[a[:, 0] > 0]
names = ["load (global -> LDS) + range reduction", "histogram", "scan of 4096 buckets", "grouping (returning adds, index stores)",
         "rank inside the bucket", "read-out (ids to memory)"]
print(f"{len(a)} workgroups stamped; mean microseconds per phase (one workgroup = one tile):")
for k, nm in enumerate(names):
    d = (a[:, k + 1] - a[:, k]) / 100.0
    print(f"  {nm:52s} mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}")
tot = (a[:, 6] - a[:, 0]) / 100.0
print(f"  whole tile: mean {tot.mean():.2f} us; kernel span {(a[:, 6].max() - a[:, 0].min()) / 100.0:.1f} us for these workgroups")
