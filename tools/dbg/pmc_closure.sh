# SQ counters of the tracking closure's kernels at 500 k (eager launches, one frame): two passes of 8 counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_c1 gpurun_out/pmc_c2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_c1 -o c1 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_c1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_c2 -o c2 -- python3 tools/prof_closure.py --frames 1 --eager > gpurun_out/pmc_c2.log 2>&1
python3 tools/dbg/show_pmc.py $(find gpurun_out/pmc_c1 -name "*counter_collection.csv") $(find gpurun_out/pmc_c2 -name "*counter_collection.csv")
