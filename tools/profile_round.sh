#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are read against (run on the GPU box through gpurun):
#   gpurun_out/bench_$TAG.json        the default bench line (headline: 500 k tracking + mapping, with its extras)
#   gpurun_out/prof_$TAG/             --kernel-trace --stats of the SAME headline loop (HIP-graph replay)
#   gpurun_out/pmc_fetch_$TAG/        --pmc FETCH_SIZE  (eager launches of the same closures / BA iterations, separate pass)
#   gpurun_out/pmc_write_$TAG/        --pmc WRITE_SIZE  (separate pass)
#   gpurun_out/pmc_sq1_$TAG/, sq2     SQ counters of the tracking closure's kernels (two passes of 8 counters, eager launches)
#   gpurun_out/prof_${TAG}_cfg5/      --kernel-trace --stats of BASELINE.json configs[4] (5 M Gaussians, SH-3, 1920x1080)
#   gpurun_out/pmc_fetch_${TAG}_cfg5/, pmc_write_${TAG}_cfg5/   its HBM traffic counters (separate passes)
# Counter passes never share a run with --stats / trace domains other than --kernel-trace.
# usage: bash tools/profile_round.sh TAG [parts: all|head|cfg5|sq] ; then python tools/distill_profiles.py TAG
set -e
TAG=${1:-r03}
PART=${2:-all}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$PART" = all ] || [ "$PART" = head ]; then
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --no-cpu-baseline --no-stage-timing --no-extras > gpurun_out/prof_$TAG.log 2>&1
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -o f -- python3 tools/prof_closure.py --frames 1 --ba 3 --eager > gpurun_out/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -o w -- python3 tools/prof_closure.py --frames 1 --ba 3 --eager > gpurun_out/pmc_write_$TAG.log 2>&1
echo "traffic counters done"
fi
if [ "$PART" = all ] || [ "$PART" = sq ]; then
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq1_$TAG -o c1 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_sq1_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq2_$TAG -o c2 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_sq2_$TAG.log 2>&1
echo "SQ counters done"
fi
if [ "$PART" = all ] || [ "$PART" = cfg5 ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_cfg5 -o ${TAG}_cfg5 -- python3 bench.py --cfg5-only > gpurun_out/prof_${TAG}_cfg5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_${TAG}_cfg5 -o f -- python3 bench.py --cfg5-only > gpurun_out/pmc_fetch_${TAG}_cfg5.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_${TAG}_cfg5 -o w -- python3 bench.py --cfg5-only > gpurun_out/pmc_write_${TAG}_cfg5.log 2>&1
echo "configs[4] done"
fi
echo profile_round done
