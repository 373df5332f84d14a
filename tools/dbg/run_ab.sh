set -e
cd $GRAFT_REPO_ROOT
timeout -k 5 60 tools/ubench/reduce_scatter_test | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "raster or fused" 2>&1 | tail -2
timeout -k 10 300 python tools/ab_raster.py 100000 500000 C=1 C=8 2>&1 | grep -E "^N=|raster_bwd vauto"
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline | cut -c1-200
