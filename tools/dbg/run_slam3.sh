set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 2>&1 | tail -1 | cut -c1-140
timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 --pose-lbfgs 2>&1 | tail -1 | cut -c1-140
timeout -k 10 300 python tools/bench_slam.py --gaussians 100000 --frames 20 2>&1 | tail -1 | cut -c1-140
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_slam -o s -- python tools/bench_slam.py --gaussians 500000 --frames 20 --pose-lbfgs > gpurun_out/prof_slam.log 2>&1
