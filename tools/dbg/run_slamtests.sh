set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_slam_loop.py tests/test_gpu_tracking.py -x -q 2>&1 | tail -5
timeout -k 10 300 python tools/run_slam.py 2>&1 | tail -3
