set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile_lists or hot_tile" 2>&1 | tail -5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "isect" 2>&1 | tail -3
timeout -k 10 300 python tools/ab_raster.py 100000 500000 C=1 C=8 2>&1 | grep -E "^N=|isect"
timeout -k 10 300 python bench.py --steps 200 --warmup 20
timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 2>&1 | tail -1
