set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ssim
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ssim -o p -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stage-timing > gpurun_out/prof_ssim.log 2>&1
tail -1 gpurun_out/prof_ssim.log | cut -c1-200
f=$(find gpurun_out/prof_ssim -name "*kernel_stats.csv" | head -1)
grep -E "ssim|map_loss|raster" $f | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-160
