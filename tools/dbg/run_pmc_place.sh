set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_pl1 -o p -- python tools/dbg/pmc_place.py 5 > gpurun_out/pmc_pl1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_pl3 -o p -- python tools/dbg/pmc_place.py 5 > gpurun_out/pmc_pl3.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_pl4 -o p -- python tools/dbg/pmc_place.py 5 > gpurun_out/pmc_pl4.log 2>&1 || true
