import sys, os, ctypes as C, runpy
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.argv = ["bench_slam.py", "--gaussians", "500000", "--frames", "20"]
try:
    runpy.run_path(os.path.join(ROOT, "tools", "bench_slam.py"), run_name="__main__")
except SystemExit:
    pass
import torch
from gslam_amd._lib import lib
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib.gsx_track_opt_dbg.restype = None
lib.gsx_track_opt_dbg(out)
n = max(1, out[5])
names = ["load+reduce", "sum12+loss+pose_bwd", "to_advance", "writeback+pose_fwd", "store"]
print("calls", out[5], {k: round(out[i] / n / 100.0, 2) for i, k in enumerate(names)}, "us (100 MHz wall clock)")
