"""bench.py with the launch plans' GSX_PROJ_SKIP_CULLED switched off (A/B of that flag)"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gslam_amd.plan as P  # noqa: E402

P._SKIP_CULLED = 0
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), run_name="__main__")
