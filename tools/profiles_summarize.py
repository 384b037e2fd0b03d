"""Turns the rocprofv3 outputs of tools/profile_round.sh (under gpurun_out/) into the summaries kept in profiles/:
python tools/profiles_summarize.py <tag>   (run in the dev container after the gpurun call has merged its outputs)"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[0]


# 1. kernel stats of the default bench run, verbatim
with open(one(f"{tag}_kt/**/*kernel_stats.csv")) as f, open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w") as o:
    o.write(f.read())
# 2. per-dispatch durations of the inflate kernel in that run (bench order: dynamic warm-up+steps, then stored, then fixed)
rows = [r for r in csv.DictReader(open(one(f"{tag}_kt/**/*kernel_trace.csv"))) if "inflate_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def _workload(i, n):  # bench order: dynamic (3 warm-up + 20 timed), then stored and fixed (each 3 + 5)
    return "dynamic" if i < 23 else ("stored" if i < 31 else "fixed") if n == 39 else ""


with open(os.path.join(P, f"{tag}_inflate_dispatches.csv"), "w") as o:
    o.write("dispatch_index,duration_ms,workload\n")
    for i, r in enumerate(rows):
        o.write(f"{i},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.4f},{_workload(i, len(rows))}\n")
# 3. PMC passes: counters per inflate dispatch; bench order with --steps 3 --warmup 1 is dynamic x4, stored x4, fixed x4
traffic = {}
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rr = [r for r in csv.DictReader(open(one(f"{tag}_pmc_{name}/**/*counter_collection.csv"))) if "inflate_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    rr.sort(key=lambda r: int(r["Dispatch_Id"]))
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
    with open(os.path.join(P, f"{tag}_pmc_{name}_size.csv"), "w") as o:
        w = csv.DictWriter(o, keep, extrasaction="ignore")
        w.writeheader()
        w.writerows(rr)
    n = len(rr) // 3
    for k, wl in enumerate(("dynamic", "stored", "fixed")):
        vals = [float(r["Counter_Value"]) for r in rr[k * n + 1:(k + 1) * n]]  # drop the warm-up dispatch
        traffic.setdefault(wl, {})[name] = sum(vals) / len(vals)
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 3 --warmup 1 --no-cpu` "
                 f"({tag}, tools/profile_round.sh); counters are in KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                 "requests at 64 B), calibrated on the stored workload whose read bytes are known; Infinity-Cache hits are included in FETCH_SIZE",
       "unit": "bytes per launch (65536 units)"}
for wl, t in traffic.items():
    rd, wr = int(t["fetch"] * 1024 * 2), int(t["write"] * 1024)
    out[wl] = {"fetch_size_kib_raw": t["fetch"], "write_size_kib": t["write"], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
# 4. the mixed (gzip + zstd) and encode workloads: kernel stats and traffic of their kernels
def _avg_counter(path_glob, kernel, ctr):
    rr = [r for r in csv.DictReader(open(one(path_glob))) if kernel in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    rr.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rr[1:]] or [float(r["Counter_Value"]) for r in rr]  # drop the warm-up dispatch
    return rr, (sum(vals) / len(vals) if vals else 0.0)


for wl, kernels in (("mixed", ("inflate_kernel", "zstd_kernel")), ("encode", ("deflate_kernel",))):
    try:
        with open(one(f"{tag}_{wl}_kt/**/*kernel_stats.csv")) as f, open(os.path.join(P, f"{tag}_{wl}_kernel_stats.csv"), "w") as o:
            o.write(f.read())
        tot = {"fetch": 0.0, "write": 0.0}
        per = {}
        for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
            allrows = []
            for k in kernels:
                rr, avg = _avg_counter(f"{tag}_{wl}_pmc_{name}/**/*counter_collection.csv", k, ctr)
                allrows += rr
                tot[name] += avg
                per.setdefault(k, {})[name] = avg
            with open(os.path.join(P, f"{tag}_{wl}_pmc_{name}_size.csv"), "w") as o:
                w = csv.DictWriter(o, keep, extrasaction="ignore")
                w.writeheader()
                w.writerows(allrows)
        rd, wr = int(tot["fetch"] * 1024 * 2), int(tot["write"] * 1024)
        out[wl] = {"fetch_size_kib_raw": tot["fetch"], "write_size_kib": tot["write"], "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                   "hbm_bytes_per_launch": rd + wr, "per_kernel_kib": per}
    except SystemExit as e:
        print("skipped", wl, e)
json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
if os.path.exists(os.path.join(G, f"{tag}_pmc_insts.txt")):
    open(os.path.join(P, f"{tag}_pmc_insts.txt"), "w").write(open(os.path.join(G, f"{tag}_pmc_insts.txt")).read())
for src, dst in ((f"{tag}_bench.json", f"{tag}_bench_rocprof_run.json"), (f"{tag}_mixed.json", f"{tag}_bench_mixed.json"), (f"{tag}_encode.json", f"{tag}_bench_encode.json")):
    if os.path.exists(os.path.join(G, src)):
        open(os.path.join(P, dst), "w").write(open(os.path.join(G, src)).read())
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items() if isinstance(v, dict)}))
