# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the BA iteration's kernels, eager launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/prof_closure.py --frames 1 --ba 3 --eager > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/prof_closure.py --frames 1 --ba 3 --eager > gpurun_out/pmc_w.log 2>&1
python3 tools/dbg/show_pmc.py $(find gpurun_out/pmc_f -name "*counter_collection.csv") $(find gpurun_out/pmc_w -name "*counter_collection.csv") | grep -A1 "${1:-ssim}"
