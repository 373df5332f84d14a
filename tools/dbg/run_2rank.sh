set -e
cd $GRAFT_REPO_ROOT
export GSX_FORCE_DEVICE=0 GSX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -4 | cut -c1-600
