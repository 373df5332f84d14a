set -e
cd $GRAFT_REPO_ROOT
for a in "50 5" "50 50" "50 500" "200 20" "1000 100" "50 5"; do set -- $a
 timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-stage-timing | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], d['ms_per_step'], d['roofline']['achieved'] if d.get('roofline') else None)"
done
