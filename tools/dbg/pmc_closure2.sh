# SQ counters of the tracking closure's kernels at 500 k (eager launches, one frame): three passes; kernel trace of the graph replay first
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x}
rm -rf gpurun_out/prof_q gpurun_out/pmc_c1 gpurun_out/pmc_c2 gpurun_out/pmc_c3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py --frames 6 > gpurun_out/prof_q.log 2>&1 || { tail -5 gpurun_out/prof_q.log; exit 1; }
python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) > gpurun_out/prof_${TAG}.txt
head -12 gpurun_out/prof_${TAG}.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_c1 -o c1 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_c1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_c2 -o c2 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_c2.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc_c3 -o c3 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_c3.log 2>&1 || tail -3 gpurun_out/pmc_c3.log
python3 tools/dbg/show_pmc.py $(find gpurun_out/pmc_c1 gpurun_out/pmc_c2 gpurun_out/pmc_c3 -name "*counter_collection.csv") > gpurun_out/pmc_${TAG}.txt
grep -A1 "fused\|front_\|tile_sort" gpurun_out/pmc_${TAG}.txt
