#!/bin/bash
# bench.py $BENCH_ARGS under builds of one source with extra compiler flags: FLAGSETS is a ';'-separated list of flag strings
# (an empty entry = the shipped flags), each run REPS times alternating
cd $GRAFT_REPO_ROOT
IFS=';' read -ra SETS <<< "$FLAGSETS"
for rep in $(seq 1 ${REPS:-2}); do
for i in "${!SETS[@]}"; do
f="${SETS[$i]}"
python3 - <<PY
from gslam_amd.csrc import build
import os, shlex
build.SOURCES["$SRC"] = list(build.SOURCES["$SRC"]) + shlex.split("""$f""")
o = os.path.join(build.OBJ, "$SRC".replace(".hip", ".o"))
if os.path.exists(o): os.remove(o)
build.build()
PY
python3 bench.py $BENCH_ARGS > gpurun_out/ab_flags_$i.json 2> gpurun_out/ab_flags_$i.err || { tail -3 gpurun_out/ab_flags_$i.err; exit 1; }
python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(repr(sys.argv[2]), d['value'], d['ms_per_step'])" gpurun_out/ab_flags_$i.json "$f"
done
done
