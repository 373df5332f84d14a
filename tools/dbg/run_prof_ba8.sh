set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba8 -o b -- python tools/dbg/prof_ba8.py > gpurun_out/prof_ba8.log 2>&1
tail -1 gpurun_out/prof_ba8.log
