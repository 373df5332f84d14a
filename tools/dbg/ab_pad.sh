cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pad gpurun_out/pmc_pl1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pad -o p -- python tools/dbg/pmc_place.py 5 > gpurun_out/prof_pad.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_pl1 -o p -- python tools/dbg/pmc_place.py 5 > gpurun_out/pmc_pl1.log 2>&1
python -c "
import csv
for r in csv.DictReader(open('gpurun_out/prof_pad/p_kernel_stats.csv')):
    if 'tile_sort_count' in r['Name'] or 'fine_place' in r['Name']: print(r['Name'][21:45], r['Calls'], float(r['AverageNs'])/1e3)
acc={}
for r in csv.DictReader(open('gpurun_out/pmc_pl1/p_counter_collection.csv')):
    if 'tile_sort_count' in r['Kernel_Name']: acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
print({k:sum(v)/len(v) for k,v in acc.items()})
"
