#!/bin/bash
# G-rank (default two; at most 4 here: a box allows 6 GPU processes) rehearsal of the keyframe-sharded BA bench on the ONE GPU of a gpurun box (both ranks on cuda:0, gloo between them;
# the driver's scaling runs use one rank per GPU over RCCL).  Checks that `bench.py --gpus $G` runs green; its numbers say
# nothing about xGMI.
cd $GRAFT_REPO_ROOT
export GSX_FORCE_DEVICE=0 GSX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-500000}
G=${2:-2}
# further arguments go to bench.py (e.g. --exchange-ab, --exchange-ranges 4)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $G --steps 10 --warmup 3 --gaussians $N ${@:3} > gpurun_out/r05_scale$G.json 2> gpurun_out/r05_scale$G.err
echo rc=$?
tail -c 2500 gpurun_out/r05_scale$G.json
grep -v "amdgpu.ids\|Gloo" gpurun_out/r05_scale$G.err | tail -8
