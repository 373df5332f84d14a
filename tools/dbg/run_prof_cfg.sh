set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
GSX_BIN_PRESORT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "isect or hot_tile or tile_lists" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -o c5 -- python tools/dbg/prof_cfg5.py 5 > gpurun_out/prof_cfg5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg4 -o c4 -- python tools/dbg/prof_cfg5.py 4 > gpurun_out/prof_cfg4.log 2>&1
tail -1 gpurun_out/prof_cfg5.log; tail -1 gpurun_out/prof_cfg4.log
