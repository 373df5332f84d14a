import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
for k, d in acc.items():
    if len(next(iter(d.values()))) < int(__import__("os").environ.get("MINCALLS", "20")):
        continue
    print(k)
    print("   " + "  ".join(f"{c}={sum(d[c]) / len(d[c]):.3g}" for c in names if c in d))
