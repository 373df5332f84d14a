set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 2>&1 | tail -1 | cut -c60-130
timeout -k 10 300 python tools/bench_slam.py --gaussians 100000 --frames 20 2>&1 | tail -1 | cut -c60-130
