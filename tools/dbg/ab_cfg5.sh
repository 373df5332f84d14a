#!/bin/bash
# A/B of a compile-time constant of isect_bin.hip on configs[4] (and the 2 M x 8 BA with BA2M=1): rebuilds with -D$DEFINE=v for each v in VALUES
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in $VALUES; do
python3 - <<PY
from gslam_amd.csrc import build
import os
build.SOURCES["isect_bin.hip"] = [f for f in build.SOURCES["isect_bin.hip"] if not f.startswith("-D$DEFINE")] + ["-D$DEFINE=$v"] + "$EXTRA".split()
o = os.path.join(build.OBJ, "isect_bin.o")
if os.path.exists(o): os.remove(o)
build.build()
PY
rm -rf gpurun_out/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o c5 -- python3 bench.py --cfg5-only > gpurun_out/prof_c5.log 2>&1 || { tail -5 gpurun_out/prof_c5.log; exit 1; }
echo "== $DEFINE=$v $EXTRA"
python3 tools/show_stats.py $(find gpurun_out/prof_c5 -name '*kernel_stats.csv' | head -1) > gpurun_out/prof_c5.txt
grep -E "tile_sort|fine_place" gpurun_out/prof_c5.txt
grep fwd_ms gpurun_out/prof_c5.log | cut -c160-230
done
