"""Copies the judged summaries of a tools/profile_round.sh run from gpurun_out/ into profiles/ (tracked):
kernel stats csv, per-kernel HBM traffic from the two PMC passes (corrected as MI355X_MICROARCH.md §HBM prescribes:
FETCH_SIZE x2 for wide coalesced reads, unit KB), the bench line, and profiles/traffic_<round>.json for bench.py.
usage: python tools/distill_profiles.py TAG [dominant-kernel-substring]"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dom = sys.argv[2] if len(sys.argv) > 2 else None       # default: the device kernel behind the bench line's roofline.kernel
g = lambda *p: os.path.join(ROOT, "gpurun_out", *p)
out = lambda n: os.path.join(ROOT, "profiles", n)

shutil.copy(g(f"prof_{tag}", f"{tag}_kernel_stats.csv"), out(f"{tag}_kernel_stats.csv"))
with open(g(f"bench_{tag}.json")) as f:
    line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
with open(out(f"{tag}_bench.json"), "w") as f:
    f.write(line + "\n")


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(g(f"pmc_fetch_{tag}", "f_counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(g(f"pmc_write_{tag}", "w_counter_collection.csv"), "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
    rows.append((k, fk, wk, (2.0 * fk + wk) * 1024.0))
rows.sort(key=lambda r: -r[3])
with open(out(f"{tag}_pmc_hbm_traffic.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "FETCH_SIZE_KB_avg_per_launch", "WRITE_SIZE_KB_avg_per_launch",
                "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
    for r in rows:
        w.writerow([r[0][:120], f"{r[1]:.1f}", f"{r[2]:.1f}", f"{r[3]:.0f}"])
bench = json.loads(line)
if dom is None:
    dom = {"gsx_raster_track_fused": "raster_track_fused", "gsx_raster_bwd": "raster_bwd",
           "gsx_raster_fwd_track_loss": "raster_fwd"}.get(bench.get("roofline", {}).get("kernel", ""), "raster_bwd")
hit = [r for r in rows if dom in r[0] and "4q<4" in r[0]] or [r for r in rows if dom in r[0]]
if hit:
    k, fk, wk, b = hit[0]
    tj = {"workload_gaussians": bench["config"]["gaussians"], "stage": bench["roofline"]["kernel"], "device_kernel": k[:100],
          "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": int(b),
          "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_closure.py --eager "
                    "(the tracking closures of one frame + 3 BA iterations, launched eagerly so that counters attribute per kernel); "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports half of wide coalesced reads); unit KB"}
    rnd = tag.split("_")[0]
    with open(out(f"traffic_{rnd}.json"), "w") as f:
        json.dump(tj, f, indent=1)
    print("dominant:", k[:80], "bytes/launch", int(b))
print("wrote", [n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.startswith(tag)])
