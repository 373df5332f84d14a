# memory-path counters of the tracking closure's front kernels at 500 k (eager launches, one frame): separate passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x}
rm -rf gpurun_out/pmc_f1 gpurun_out/pmc_f2 gpurun_out/pmc_f3 gpurun_out/pmc_f4
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/pmc_f1 -o c1 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_f1.log 2>&1 || tail -3 gpurun_out/pmc_f1.log
# (this pass hung rocprofv3 on the pool - killed after 7 silent minutes; left here as a warning, do not run)
# rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_f2 -o c2 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_f2.log 2>&1 || tail -3 gpurun_out/pmc_f2.log
# rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum --kernel-trace --output-format csv -d gpurun_out/pmc_f3 -o c3 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_f3.log 2>&1 || tail -3 gpurun_out/pmc_f3.log
# rocprofv3 --pmc TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_f4 -o c4 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_f4.log 2>&1 || tail -3 gpurun_out/pmc_f4.log
MINCALLS=10 python3 tools/dbg/show_pmc.py $(find gpurun_out/pmc_f1 -name "*counter_collection.csv") > gpurun_out/pmc_front_${TAG}.txt
grep -A1 "front_\|fused\|column" gpurun_out/pmc_front_${TAG}.txt
