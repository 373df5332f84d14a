#!/bin/bash
# same-box A/B of bench.py with a python one-liner applied before it runs: PATCH_A / PATCH_B (python statements; `P` = gslam_amd.plan)
cd $GRAFT_REPO_ROOT
for v in A B A B; do
  eval "patch=\$PATCH_$v"
  python3 - <<PY > gpurun_out/ab_env_$v.json 2> gpurun_out/ab_env_$v.err || { tail -3 gpurun_out/ab_env_$v.err; exit 1; }
import sys, runpy
sys.path.insert(0, '.')
import gslam_amd.plan as P
$patch
sys.argv = ['bench.py'] + '''$BENCH_ARGS'''.split()
runpy.run_path('bench.py', run_name='__main__')
PY
  python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d.get('closure', {}).get('us'), d.get('ba_iteration', {}).get('us'), d.get('stage_us'))" gpurun_out/ab_env_$v.json
done
