#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are read against (run on the GPU box through gpurun):
#   gpurun_out/prof_$TAG/       --kernel-trace --stats of the default bench (HIP-graph replay)
#   gpurun_out/pmc_fetch_$TAG/  --pmc FETCH_SIZE  (eager, separate pass)
#   gpurun_out/pmc_write_$TAG/  --pmc WRITE_SIZE  (eager, separate pass)
# usage: bash tools/profile_round.sh TAG
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --steps 200 --warmup 20 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage-timing > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -o f -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-timing --no-graph > gpurun_out/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -o w -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-timing --no-graph > gpurun_out/pmc_write_$TAG.log 2>&1
