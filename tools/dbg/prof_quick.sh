#!/bin/bash
# kernel-trace of the tracking closures only (graph replay), summary printed; ARGS = extra prof_closure.py arguments
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for variant in "${VARIANTS[@]:-}"; do :; done
run() {
  rm -rf gpurun_out/prof_q
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py --frames ${FRAMES:-6} $1 > gpurun_out/prof_q.log 2>&1 || { tail -5 gpurun_out/prof_q.log; exit 1; }
  echo "== prof_closure $1"
  python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) > gpurun_out/prof_q.txt
  head -${LINES_N:-9} gpurun_out/prof_q.txt
}
run "$A_ARGS"
if [ -n "$B_ARGS" ]; then run "$B_ARGS"; fi
