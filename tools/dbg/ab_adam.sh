cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stage-timing > gpurun_out/prof_q.log 2>&1
python -c "
import csv
for r in csv.DictReader(open('gpurun_out/prof_q/q_kernel_stats.csv')):
    if 'adam_multi' in r['Name']: print('adam', r['Calls'], float(r['AverageNs'])/1e3)
"
tail -1 gpurun_out/prof_q.log | cut -c1-160
