#!/bin/bash
# A/B of the candidate margins of the tracking closure inside the headline bench (tuning runs; lines are diagnostic)
cd $GRAFT_REPO_ROOT
for m in "$@"; do
  python3 bench.py --no-extras --no-cpu-baseline --no-stage-timing --diag cand:$m > gpurun_out/r03_c_$m.json 2> gpurun_out/r03_c_$m.err
  python3 - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r03_c_$m.json").read().splitlines() if x.startswith("{")][-1])
print("$m", d["value"], d["closure"]["us"], d["ba_iteration"]["us"], d["config"].get("tracking_candidates"), flush=True)
PY
done
