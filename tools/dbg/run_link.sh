set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline --no-stage-timing | cut -c1-200
rm -rf gpurun_out/prof_ssim
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ssim -o p -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stage-timing > gpurun_out/prof_ssim.log 2>&1
grep -E "pose|finish" gpurun_out/prof_ssim/p_kernel_stats.csv | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-160
timeout -k 10 300 python tools/bench_slam.py --gaussians 500000 --frames 20 2>&1 | tail -1 | cut -c1-140
