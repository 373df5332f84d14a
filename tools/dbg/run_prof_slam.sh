set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/tile_stats.py > gpurun_out/tile_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_slam500k -o slam -- python tools/bench_slam.py --gaussians 500000 --frames 5 > gpurun_out/prof_slam500k.log 2>&1
