set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "adam or ba_step or optim or decay or fused_ba or graph" 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline --no-stage-timing | cut -c1-200
rm -rf gpurun_out/prof_ssim
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ssim -o p -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stage-timing > gpurun_out/prof_ssim.log 2>&1
grep -E "adam" gpurun_out/prof_ssim/p_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-160
