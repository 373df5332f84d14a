set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile_lists or hot_tile" 2>&1 | tail -5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "isect" 2>&1 | tail -3
timeout -k 10 300 python tools/bench_configs.py
