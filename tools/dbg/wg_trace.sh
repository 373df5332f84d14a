#!/bin/bash
# builds the DIAGNOSTIC library (raster.hip with -DGSX_WG_TRACE) over the box's scratch copy of libgsx.so and runs wg_trace.py
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
from gslam_amd.csrc import build
build.SOURCES["raster.hip"] = build.SOURCES["raster.hip"] + ["-DGSX_WG_TRACE"]
import os
os.remove(os.path.join(build.OBJ, "raster.o"))
build.build()
PY
python3 tools/dbg/wg_trace.py $WG_ARGS > gpurun_out/wg_trace.txt 2>&1 || { tail -20 gpurun_out/wg_trace.txt; exit 1; }
cat gpurun_out/wg_trace.txt
