cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "" "--no-balance"; do
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py --frames 6 $a > gpurun_out/prof_q.log 2>&1 || { tail -5 gpurun_out/prof_q.log; exit 1; }
echo "== args: $a"
python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) | head -7
done
