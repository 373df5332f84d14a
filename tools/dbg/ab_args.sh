# kernel trace of prof_closure.py for every argument string in ARGS_LIST (separated by ';'); prints the kernels matching $KERNELS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IFS=';' read -ra LIST <<< "$ARGS_LIST"
for a in "${LIST[@]}"; do
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py --frames ${FRAMES:-6} $a > gpurun_out/prof_q.log 2>&1 || { tail -5 gpurun_out/prof_q.log; exit 1; }
python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) > gpurun_out/prof_q.txt
echo "== $a : $(grep -E "${KERNELS:-fused}" gpurun_out/prof_q.txt | cut -c72-140)"
done
