set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_slam -o s -- python tools/bench_slam.py --gaussians 500000 --frames 20 > gpurun_out/prof_slam.log 2>&1
tail -2 gpurun_out/prof_slam.log
