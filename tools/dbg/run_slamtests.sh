set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_slam_loop.py tests/test_gpu_tracking.py -x -q 2>&1 | tail -3
for v in fused split fused split; do
echo -n "$v: "; GSX_TRACK_TAIL=$v timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 2>&1 | tail -1 | cut -c60-130
done
