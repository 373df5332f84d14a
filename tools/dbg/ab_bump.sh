cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 1 0; do
  echo -n "GSX_ADAM_SELF_BUMP=$v: "
  GSX_ADAM_SELF_BUMP=$v timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_us'].get('gsx_adam_multi_steps'), d['stage_us'])"
done
done
