set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python tools/bench_configs.py
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline | cut -c1-200
