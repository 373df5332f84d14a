set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 2>&1 | tail -1 | cut -c1-140
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_slam -o s -- python tools/bench_slam.py --gaussians 500000 --frames 20 > gpurun_out/prof_slam.log 2>&1
grep -E "project_fwd|project_bwd_kernel<true>" gpurun_out/prof_slam/s_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-140
