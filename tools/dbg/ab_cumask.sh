#!/bin/bash
# headline with the mapping stream restricted to a CU pattern (bench.py --diag cumask:<hex>), same box
cd $GRAFT_REPO_ROOT
for m in ${MASKS:-none 77777777 none 33333333 55555555}; do
  if [ "$m" = none ]; then d=""; else d="--diag cumask:$m"; fi
  python3 bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-stage-timing $d > gpurun_out/ab_cumask.json 2> gpurun_out/ab_cumask.err || { tail -3 gpurun_out/ab_cumask.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_cumask.json').read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d.get('INVALID','')[:40])"
done
