import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.bench_configs as B
B.run("3: 500k, 640x480, 8 cameras (BA window)", 500_000, 8, 640, 480)
