#!/bin/bash
# A/B of the tight tile lists (RenderPlan.TIGHT_LISTS) on the BA iteration (500 k x 8) and the window refiner's closure
cd $GRAFT_REPO_ROOT
for t in ${MODES:-1 0}; do
  echo "== mode $t (0: means2d + radii, 1: tight rectangles from the projection, 2: the reference rectangles from the projection)"
  GSX_TIGHT=$t timeout -k 10 300 python3 - <<PY
import os, sys, json, time, torch
sys.path.insert(0, '.')
import gslam_amd.plan as P
P.RenderPlan.TIGHT_LISTS = os.environ['GSX_TIGHT'] == '1'
P.RenderPlan.RECT_LISTS = os.environ['GSX_TIGHT'] == '2'
import bench
dev = torch.device('cuda:0')
r = bench.run_ba(dev, 0, 1, int(os.environ.get('N', 500000)), 640, 480, 8, 60, 10, stage_timing=True)
print('BA %d x 8: %.1f us per iteration, M=%d' % (int(os.environ.get('N', 500000)), r['ms_per_iter'] * 1e3, r['n_isects_local']))
print({k: v for k, v in r.get('stage_us', {}).items()})
PY
  GSX_TIGHT=$t timeout -k 10 300 python3 - <<PY
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/dbg')
import gslam_amd.plan as P
P.RenderPlan.TIGHT_LISTS = os.environ['GSX_TIGHT'] == '1'
P.RenderPlan.RECT_LISTS = os.environ['GSX_TIGHT'] == '2'
import window_closure_run
window_closure_run.main()
PY
done
