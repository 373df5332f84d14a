#!/bin/bash
# kernel stats of the configs[3] workload on one GPU (2 M Gaussians, 8-keyframe window, BA iterations only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ba2m
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba2m -o b -- python3 bench.py --ba-only --no-cpu-baseline --no-stage-timing ${BA_ARGS} > gpurun_out/prof_ba2m.log 2>&1 || { tail -5 gpurun_out/prof_ba2m.log; exit 1; }
python3 tools/show_stats.py $(find gpurun_out/prof_ba2m -name '*kernel_stats.csv' | head -1) > gpurun_out/prof_ba2m.txt
head -${LINES_N:-26} gpurun_out/prof_ba2m.txt
