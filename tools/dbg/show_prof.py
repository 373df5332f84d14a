import csv, sys
for tag, f in (("cfg5", '/root/repo/gpurun_out/prof_cfg5/c5_kernel_stats.csv'), ("cfg4 C=8", '/root/repo/gpurun_out/prof_cfg4/c4_kernel_stats.csv')):
    rows = list(csv.DictReader(open(f)))
    print("==", tag)
    for r in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 12]:
        print(f"{r['Name'][:78]:78s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:9.1f}us {float(r['Percentage']):5.1f}%")
