"""prints a rocprofv3 --stats kernel csv as a table: python tools/show_stats.py path/to/X_kernel_stats.csv [n]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:70]:70s} calls={r['Calls']:>6s} avg={float(r['AverageNs']) / 1e3:8.1f}us tot={float(r['TotalDurationNs']) / 1e6:8.2f}ms {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
