# memory-path counters of the tracking closure's front kernels at 500 k (eager launches, one frame): separate passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x}
# ONE pass, at most four counters per hardware block (here: 3 TCP + 4 TCC).  Round 4 had three more passes in this file; the one that
# was run - TCC_EA0_RDREQ_LEVEL / _DRAM / TCC_TAG_STALL / TCC_REQ / TCC_READ + one TCP counter: FIVE counters of the TCC block,
# whose channels have four counter registers each - made rocprofv3 sit silent for 7 minutes until the pool killed it (rocprofv3
# does not split a --pmc list into passes).  The other two asked for six TCP_UTCL1_* counters (TCP: four registers) and were
# never run.  They are gone from the tool: split such lists into groups of <= 4 per block, one rocprofv3 run each.
rm -rf gpurun_out/pmc_f1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/pmc_f1 -o c1 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_f1.log 2>&1 || tail -3 gpurun_out/pmc_f1.log
MINCALLS=10 python3 tools/dbg/show_pmc.py $(find gpurun_out/pmc_f1 -name "*counter_collection.csv") > gpurun_out/pmc_front_${TAG}.txt
grep -A1 "front_\|fused\|column" gpurun_out/pmc_front_${TAG}.txt
