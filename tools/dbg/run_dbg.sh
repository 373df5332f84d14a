cd $GRAFT_REPO_ROOT
AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-stage-timing 2>&1 | grep -v "^$" | tail -30
