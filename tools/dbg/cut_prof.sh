# fused launch time against the cut-off margin of the in-rasteriser tile sort (kernel trace of graph-replayed closures)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ${MARGINS:-0.05 0.1 0.2 0.3 0.5}; do
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py --frames 6 --cut-margin $m > gpurun_out/prof_q.log 2>&1 || { tail -5 gpurun_out/prof_q.log; exit 1; }
echo "== margin $m: $(grep 'tile sort' gpurun_out/prof_q.log | cut -c1-140)"
python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) | grep fused
done
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py --frames 6 --no-defer-sort > gpurun_out/prof_q.log 2>&1
echo "== stand-alone sort launch"
python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) | grep "fused\|tile_sort"
