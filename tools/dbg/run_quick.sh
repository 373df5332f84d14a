set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ba_step.py tests/test_gpu_parity.py -x -q -k "isect or tile or ba or binning" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-stage-timing > gpurun_out/prof_q.log 2>&1
grep -E "tile_scan|column_scan|count_matrix" gpurun_out/prof_q/q_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
