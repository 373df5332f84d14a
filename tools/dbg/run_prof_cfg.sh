set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -o c5 -- python tools/dbg/prof_cfg5.py 5 > gpurun_out/prof_cfg5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg4 -o c4 -- python tools/dbg/prof_cfg5.py 4 > gpurun_out/prof_cfg4.log 2>&1
