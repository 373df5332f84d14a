set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -o sq1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timing --no-graph > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o sq2 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timing --no-graph > gpurun_out/pmc_sq2.log 2>&1
