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

def find(root, suffix):
    """rocprofv3 nests its output under <dir>/<host>/...: first file below ``root`` that ends with ``suffix``"""
    for d, _, files in os.walk(root):
        for f in sorted(files):
            if f.endswith(suffix):
                return os.path.join(d, f)
    return None


shutil.copy(find(g(f"prof_{tag}"), "kernel_stats.csv"), out(f"{tag}_kernel_stats.csv"))
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


def traffic_table(fetch_dir, write_dir, dst):
    fetch = per_kernel(find(fetch_dir, "counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel(find(write_dir, "counter_collection.csv"), "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        rows.append((k, fk, wk, (2.0 * fk + wk) * 1024.0))
    rows.sort(key=lambda r: -r[3])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "FETCH_SIZE_KB_avg_per_launch", "WRITE_SIZE_KB_avg_per_launch",
                    "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for r in rows:
            w.writerow([r[0][:120], f"{r[1]:.1f}", f"{r[2]:.1f}", f"{r[3]:.0f}"])
    return rows


rows = traffic_table(g(f"pmc_fetch_{tag}"), g(f"pmc_write_{tag}"), out(f"{tag}_pmc_hbm_traffic.csv"))
# BASELINE.json configs[4] (5 M Gaussians, SH-3, 1920x1080): kernel stats + HBM traffic of `bench.py --cfg5-only`
if os.path.isdir(g(f"prof_{tag}_cfg5")):
    shutil.copy(find(g(f"prof_{tag}_cfg5"), "kernel_stats.csv"), out(f"{tag}_cfg5_kernel_stats.csv"))
    if os.path.isdir(g(f"pmc_fetch_{tag}_cfg5")):
        traffic_table(g(f"pmc_fetch_{tag}_cfg5"), g(f"pmc_write_{tag}_cfg5"), out(f"{tag}_cfg5_pmc_hbm_traffic.csv"))
# SQ counters of the tracking closure's kernels (two passes): per-launch averages + what they say about the limiter
if os.path.isdir(g(f"pmc_sq1_{tag}")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (g(f"pmc_sq1_{tag}"), g(f"pmc_sq2_{tag}")):
        path = find(d, "counter_collection.csv")
        if path is None:
            continue
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    sq = {}
    for k, d in acc.items():
        if len(next(iter(d.values()))) < 20:
            continue
        m = {c: sum(v) / len(v) for c, v in d.items()}
        waves = m.get("SQ_WAVES", 0.0)
        if waves > 0:
            m["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0.0) / waves
            m["salu_insts_per_wave"] = m.get("SQ_INSTS_SALU", 0.0) / waves
            m["lds_insts_per_wave"] = m.get("SQ_INSTS_LDS", 0.0) / waves
        if m.get("SQ_WAVE_CYCLES"):
            m["wait_any_frac_of_wave_cycles"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            m["wait_inst_frac_of_wave_cycles"] = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_INSTS_VALU"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles (4 clocks) a SIMD spends issuing VALU work: ~1.05 per wave64 instruction
            m["valu_quad_cycles_per_inst"] = m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_INSTS_VALU"]
        if m.get("SQ_WAVE_CYCLES") and waves > 0:
            m["wave_lifetime_quad_cycles"] = m["SQ_WAVE_CYCLES"] / waves
            # share of a wavefront's lifetime in which it is issuing VALU work: the rest is waiting or other pipes
            m["valu_issue_frac_of_wave_lifetime"] = m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_WAVE_CYCLES"]
        sq[k.replace("(anonymous namespace)::", "")[:100]] = {c: round(v, 4) for c, v in sorted(m.items())}
    with open(out(f"{tag}_sq_counters.json"), "w") as f:
        json.dump({"method": "rocprofv3 --pmc, two passes of 8 SQ counters over tools/prof_closure.py --frames 1 --eager (the 36 "
                             "tracking closures of one frame at 500 k Gaussians, launched eagerly so that the counters attribute per "
                             "kernel); averages per launch; SQ_ACTIVE_INST_VALU is summed over the 4 SIMDs of a CU",
                   "kernels": sq}, f, indent=1)
bench = json.loads(line)
if dom is None:
    dom = {"gsx_raster_track_fused": "raster_track_fused", "gsx_raster_track_fused_sorting": "raster_track_fused", "gsx_raster_track_fused_rows": "raster_track_fused", "gsx_raster_bwd": "raster_bwd",
           "gsx_raster_fwd_track_loss": "raster_fwd"}.get(bench.get("roofline", {}).get("kernel", ""), "raster_bwd")
hit = [r for r in rows if dom in r[0] and "4q<4" in r[0]] or [r for r in rows if dom in r[0]]
if hit:
    k, fk, wk, b = hit[0]
    tj = {"workload_gaussians": bench["config"]["gaussians"], "stage": bench["roofline"]["kernel"], "device_kernel": k[:100],
          "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": int(b),
          "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_closure.py --eager "
                    "(the tracking closures of one frame + 3 BA iterations, launched eagerly so that counters attribute per kernel); "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports half of wide coalesced reads); unit KB"}
    # the rocprofv3 kernel-trace average of the same device kernel inside the bench's own loop (graph replay): the figure
    # bench.py's roofline carries as avg_launch_us_trace, so that the line's fraction can be recomputed from profiles/
    with open(out(f"{tag}_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            if dom in r["Name"]:
                tj["trace_avg_launch_us"] = round(float(r["AverageNs"]) / 1e3, 2)
                tj["trace_calls"] = int(r["Calls"])
                tj["trace_source"] = f"profiles/{tag}_kernel_stats.csv"
                break
    rnd = tag.split("_")[0]
    sqp = out(f"{tag}_sq_counters.json")
    if os.path.exists(sqp):
        kern = json.load(open(sqp))["kernels"]
        hit_sq = [v for kk, v in kern.items() if dom in kk]
        if hit_sq:
            v = hit_sq[0]
            tj["limiter"] = {
                "kind": "VALU issue / dependent-issue latency (not HBM: see the counter traffic)",
                "valu_insts_per_wave": v.get("valu_insts_per_wave"),
                "valu_quad_cycles_per_inst": v.get("valu_quad_cycles_per_inst"),
                "valu_issue_frac_of_wave_lifetime": v.get("valu_issue_frac_of_wave_lifetime"),
                "wait_any_frac_of_wave_cycles": v.get("wait_any_frac_of_wave_cycles"),
                "lds_insts_per_wave": v.get("lds_insts_per_wave"), "source": f"profiles/{tag}_sq_counters.json"}
    with open(out(f"traffic_{rnd}.json"), "w") as f:
        json.dump(tj, f, indent=1)
    print("dominant:", k[:80], "bytes/launch", int(b))
print("wrote", [n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.startswith(tag)])
