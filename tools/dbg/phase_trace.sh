#!/bin/bash
# builds the DIAGNOSTIC library (raster.hip with -DGSX_WG_TRACE $EXTRA_DEFS) over the box's scratch copy of libgsx.so and runs
# phase_trace.py; EXTRA_DEFS e.g. "-DGSX_DBG_HALF=1"
set -e
cd $GRAFT_REPO_ROOT
python3 - <<PY
from gslam_amd.csrc import build
build.SOURCES["raster.hip"] = build.SOURCES["raster.hip"] + ["-DGSX_WG_TRACE"] + "${EXTRA_DEFS}".split()
import os
os.remove(os.path.join(build.OBJ, "raster.o"))
build.build()
PY
python3 tools/dbg/phase_trace.py > gpurun_out/phase_trace${TAG}.txt 2>&1 || { tail -20 gpurun_out/phase_trace${TAG}.txt; exit 1; }
cat gpurun_out/phase_trace${TAG}.txt
