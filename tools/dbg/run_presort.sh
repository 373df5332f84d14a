set -e
cd $GRAFT_REPO_ROOT
GSX_BIN_PRESORT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "isect or hot_tile or tile_lists" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python tools/bench_configs.py
