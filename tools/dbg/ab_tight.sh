#!/bin/bash
# A/B of the tight tile lists (RenderPlan.TIGHT_LISTS) on the BA iteration (500 k x 8) and the window refiner's closure
cd $GRAFT_REPO_ROOT
for t in 1 0; do
  echo "== TIGHT_LISTS=$t"
  GSX_TIGHT=$t timeout -k 10 300 python3 - <<PY
import os, sys, json, time, torch
sys.path.insert(0, '.')
import gslam_amd.plan as P
P.RenderPlan.TIGHT_LISTS = bool(int(os.environ['GSX_TIGHT']))
import bench
dev = torch.device('cuda:0')
r = bench.run_ba(dev, 0, 1, 500000, 640, 480, 8, 60, 10, stage_timing=True)
print('BA 500k x 8: %.1f us per iteration, M=%d' % (r['ms_per_iter'] * 1e3, r['n_isects_local']))
print({k: v for k, v in r.get('stage_us', {}).items()})
PY
  GSX_TIGHT=$t timeout -k 10 300 python3 - <<PY
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/dbg')
import gslam_amd.plan as P
P.RenderPlan.TIGHT_LISTS = bool(int(os.environ['GSX_TIGHT']))
import window_closure_run
window_closure_run.main()
PY
done
