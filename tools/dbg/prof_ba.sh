# kernel trace of 30 graph-replayed BA iterations at 500 k x 8 (after 1 frame of tracking)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py --frames 1 --ba 30 > gpurun_out/prof_q.log 2>&1 || { tail -5 gpurun_out/prof_q.log; exit 1; }
python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) 40 | grep -v "calls=    3[6-9] \|calls=     [0-9] \|calls=    [0-2][0-9] " | head -30
