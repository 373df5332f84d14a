import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.bench_configs as B
which = sys.argv[1] if len(sys.argv) > 1 else "5"
if which == "5":
    B.run("5: 5M SH-3, 1920x1080, 1 camera", 5_000_000, 1, 1920, 1080, sh=True)
else:
    B.run("4: 2M, 640x480, 8 cameras (unsharded)", 2_000_000, 8, 640, 480)
