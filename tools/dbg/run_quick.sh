set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba8 -o b -- python tools/dbg/prof_ba8.py > gpurun_out/prof_ba8.log 2>&1
tail -1 gpurun_out/prof_ba8.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline | cut -c1-200
