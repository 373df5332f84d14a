#!/bin/bash
# bench.py $BENCH_ARGS under two builds of one source (-D$DEFINE=v for v in $VALUES)
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
python3 bench.py $BENCH_ARGS > gpurun_out/ab_bench_$v.json 2> gpurun_out/ab_bench_$v.err || { tail -3 gpurun_out/ab_bench_$v.err; exit 1; }
python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('$DEFINE=$v', d['value'], d['ms_per_step'])" gpurun_out/ab_bench_$v.json
done
