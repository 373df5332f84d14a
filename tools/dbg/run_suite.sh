set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout -k 10 300 python bench.py --steps 200 --warmup 20 | cut -c1-330
timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 2>&1 | tail -1
