set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_ba_step.py -x -q -k "isect or hot_tile or tile_lists or binning or sync_free" 2>&1 | tail -3
bash tools/dbg/ab_pad.sh
timeout -k 10 300 python tools/bench_configs.py 2>&1 | tail -3
