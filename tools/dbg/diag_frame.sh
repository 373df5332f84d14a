#!/bin/bash
# where does the frame time go beyond the 36 closures: headline with the BA rounds and / or the output render left out
cd $GRAFT_REPO_ROOT
for d in ${DIAGS:-full no-ba no-out no-ba,no-out}; do
  python3 bench.py --no-cpu-baseline --no-stage-timing --no-extras --diag "${d/full/}" > gpurun_out/diag_"$d".json 2>> gpurun_out/diag.err || { tail -5 gpurun_out/diag.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['closure']['us'], d['ba_iteration']['us'])" gpurun_out/diag_"$d".json
done
