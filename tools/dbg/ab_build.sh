#!/bin/bash
# A/B of a compile-time constant on the GPU box: rebuilds one source with -D$DEFINE for each value in VALUES and profiles
# prof_closure.py ($ARGS); prints the kernels matching $KERNELS
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in $VALUES; do
python3 - <<PY
from gslam_amd.csrc import build
import os
build.SOURCES["$SRC"] = [f for f in build.SOURCES["$SRC"] if not f.startswith("-D$DEFINE")] + ["-D$DEFINE=$v"]
o = os.path.join(build.OBJ, "$SRC".replace(".hip", ".o"))
if os.path.exists(o): os.remove(o)
build.build()
PY
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 tools/prof_closure.py $ARGS > gpurun_out/prof_q.log 2>&1 || { tail -5 gpurun_out/prof_q.log; exit 1; }
echo "== $DEFINE=$v"
python3 tools/show_stats.py $(find gpurun_out/prof_q -name '*kernel_stats.csv' | head -1) > gpurun_out/prof_q.txt
grep -E "$KERNELS" gpurun_out/prof_q.txt
done
