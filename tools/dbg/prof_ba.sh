cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ba
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba -o p -- python3 tools/prof_closure.py --frames 1 --ba 20 $BA_ARGS > gpurun_out/prof_ba.log 2>&1
python tools/show_stats.py $(find gpurun_out/prof_ba -name "p_kernel_stats.csv") 24
