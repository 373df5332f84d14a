#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are read against (run on the GPU box through gpurun):
#   gpurun_out/bench_$TAG.json     the default bench line (headline: 500 k tracking + mapping)
#   gpurun_out/prof_$TAG/          --kernel-trace --stats of the SAME command (HIP-graph replay)
#   gpurun_out/pmc_fetch_$TAG/     --pmc FETCH_SIZE  (eager launches of the same closures / BA iterations, separate pass)
#   gpurun_out/pmc_write_$TAG/     --pmc WRITE_SIZE  (separate pass)
# usage: bash tools/profile_round.sh TAG ; then python tools/distill_profiles.py TAG
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --no-cpu-baseline --no-stage-timing --no-extras > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -o f -- python3 tools/prof_closure.py --frames 1 --ba 3 --eager > gpurun_out/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -o w -- python3 tools/prof_closure.py --frames 1 --ba 3 --eager > gpurun_out/pmc_write_$TAG.log 2>&1
echo profile_round done
