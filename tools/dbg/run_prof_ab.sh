set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -o ab -- python tools/ab_raster.py 500000 C=8 > gpurun_out/prof_ab.log 2>&1
