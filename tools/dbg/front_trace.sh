#!/bin/bash
# builds the DIAGNOSTIC library (isect_bin.hip with -DGSX_WG_TRACE) over the box's scratch copy of libgsx.so and runs front_trace.py
set -e
cd $GRAFT_REPO_ROOT
python3 - <<PY
from gslam_amd.csrc import build
build.SOURCES["isect_bin.hip"] = build.SOURCES["isect_bin.hip"] + ["-DGSX_WG_TRACE"] + "${EXTRA_DEFS}".split()
import os
os.remove(os.path.join(build.OBJ, "isect_bin.o"))
build.build()
PY
python3 tools/dbg/front_trace.py > gpurun_out/front_trace${TAG}.txt 2>&1 || { tail -20 gpurun_out/front_trace${TAG}.txt; exit 1; }
cat gpurun_out/front_trace${TAG}.txt
