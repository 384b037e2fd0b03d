"""Turns the rocprofv3 outputs of tools/profile_round.sh (under gpurun_out/) into the summaries kept in profiles/:
python tools/profiles_summarize.py <tag>   (run in the dev container after the gpurun call has merged its outputs)"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[0]


# 1. kernel stats of the default bench run, verbatim
with open(one(f"{tag}_kt/**/*kernel_stats.csv")) as f, open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w") as o:
    o.write(f.read())
# 2. per-dispatch durations of the codec kernels in that run, in launch order.  bench.py's default run: dynamic (3 warm-up + 20
# timed), stored, fixed (3 + 5 each), mixed (route + inflate + zstd per step, 3 + 5), encode (3 + 5, then one inflate that verifies it)
rows = [r for r in csv.DictReader(open(one(f"{tag}_kt/**/*kernel_trace.csv"))) if any(k in r["Kernel_Name"] for k in ("inflate_kernel", "zstd_kernel", "deflate_kernel", "route_kernel"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
plan = [("dynamic", 23), ("stored", 8), ("fixed", 8)]
with open(os.path.join(P, f"{tag}_dispatches.csv"), "w") as o:
    o.write("dispatch_index,kernel,duration_ms,workload\n")
    n_inf = 0
    for i, r in enumerate(rows):
        short = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        wl = ""
        if "inflate_kernel" in short:
            acc = 0
            for name, cnt in plan:
                if acc <= n_inf < acc + cnt:
                    wl = name
                acc += cnt
            if not wl:
                wl = "mixed (gzip half)" if n_inf < acc + 8 else "encode verification"
            n_inf += 1
        elif "zstd_kernel" in short or "route_kernel" in short:
            wl = "mixed"
        elif "deflate" in short:
            wl = "encode"
        o.write(f"{i},{short},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.4f},{wl}\n")
# 3. PMC passes per workload: FETCH_SIZE / WRITE_SIZE per dispatch of the workload's kernels (the first dispatch is the warm-up)
import subprocess

try:
    build = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
except Exception:
    build = None
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python3 bench.py --workload W --steps 3 --warmup 1 --no-cpu "
                 f"--extra 0` ({tag}, tools/profile_round.sh); counters are in KiB.  Rule per access shape (measured with tools/exp/fetch_probe.hip, "
                 f"profiles/{tag}_fetch_probe.txt): FETCH_SIZE reports HALF the bytes of lane-contiguous 16-byte loads (8.0 B per load) and the 64-byte "
                 "request of a scattered 16-byte load as it is (63.9 B per aligned load, 71.4 B at any alignment).  So the LZ77 source fetches of the "
                 "inflate kernel -- FETCH_SIZE of the product minus FETCH_SIZE of the -DCHIP_EXP_NO_GLOB build that leaves them out -- are counted "
                 "once and the rest of the read side twice; the level-1 encoder's candidate gathers likewise (-DCHIP_EXP_NO_CAND).  Workloads without such "
                 "an ablation (stored, mixed) keep the plain doubling, which overstates mixed's gathers.  Infinity-Cache hits are included in FETCH_SIZE",
       "build": build,
       "unit": "bytes per launch (65536 units)"}
KERNELS = {"dynamic": ("inflate_kernel",), "stored": ("inflate_kernel",), "fixed": ("inflate_kernel",), "mixed": ("inflate_kernel", "zstd_kernel"), "encode": ("deflate_kernel",)}
keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
for wl, kernels in KERNELS.items():
    try:
        tot, per = {"fetch": 0.0, "write": 0.0}, {}
        for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            allrows = []
            for k in kernels:
                rr = [r for r in csv.DictReader(open(one(f"{tag}_{wl}_pmc_{name}/**/*counter_collection.csv"))) if k in r["Kernel_Name"] and r["Counter_Name"] == ctr]
                rr.sort(key=lambda r: int(r["Dispatch_Id"]))
                if wl == "encode":
                    rr = rr[:4]  # (the run's last inflate dispatch verifies the encoder's output: not part of the workload)
                vals = [float(r["Counter_Value"]) for r in rr[1:4]] or [float(r["Counter_Value"]) for r in rr]
                avg = sum(vals) / len(vals) if vals else 0.0
                allrows += rr
                tot[name] += avg
                per.setdefault(k, {})[name] = avg
            with open(os.path.join(P, f"{tag}_{wl}_pmc_{name}_size.csv"), "w") as o:
                w = csv.DictWriter(o, keep, extrasaction="ignore")
                w.writeheader()
                w.writerows(allrows)
        scat = None
        try:  # the ablation pass of this workload, when it was taken
            rr = [r for r in csv.DictReader(open(one(f"{tag}_{wl}_noglob_pmc_fetch/**/*counter_collection.csv"))) if any(k in r["Kernel_Name"] for k in kernels) and r["Counter_Name"] == "FETCH_SIZE"]
            rr.sort(key=lambda r: int(r["Dispatch_Id"]))
            if wl == "encode":
                rr = rr[:4]
            vals = [float(r["Counter_Value"]) for r in rr[1:4]] or [float(r["Counter_Value"]) for r in rr]
            scat = max(0.0, tot["fetch"] - sum(vals) / len(vals))
        except (SystemExit, ZeroDivisionError):
            pass
        rd = int((tot["fetch"] - (scat or 0.0)) * 1024 * 2 + (scat or 0.0) * 1024)
        wr = int(tot["write"] * 1024)
        out[wl] = {"fetch_size_kib_raw": tot["fetch"], "write_size_kib": tot["write"], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
        if scat is not None:
            out[wl]["source_fetch_kib_raw"] = scat
        if len(kernels) > 1:
            out[wl]["per_kernel_kib"] = per
    except SystemExit as e:
        print("skipped", wl, e)
json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
if os.path.exists(os.path.join(G, f"{tag}_pmc_insts.txt")):
    open(os.path.join(P, f"{tag}_pmc_insts.txt"), "w").write(open(os.path.join(G, f"{tag}_pmc_insts.txt")).read())
try:  # the access-shape probe: bytes FETCH_SIZE reports per 16-byte load
    import re
    log = open(os.path.join(G, f"{tag}_fetch_probe.log")).read()
    loads = float(re.search(r"loads per kernel: (\d+)", log).group(1))
    lines = [f"tools/exp/fetch_probe.hip under rocprofv3 --pmc FETCH_SIZE ({tag}); {loads:.0f} 16-byte loads per kernel over a 2 GiB buffer"]
    for r in csv.DictReader(open(one(f"{tag}_fetch_probe/**/*counter_collection.csv"))):
        if r["Counter_Name"] == "FETCH_SIZE" and "probe" in r["Kernel_Name"]:
            lines.append(f"{r['Kernel_Name'][:60]:60s} FETCH_SIZE {float(r['Counter_Value']):12.0f} KiB = {float(r['Counter_Value']) * 1024 / loads:6.2f} B per load")
    open(os.path.join(P, f"{tag}_fetch_probe.txt"), "w").write("\n".join(lines) + "\n")
except (SystemExit, OSError, AttributeError) as e:
    print("fetch probe skipped", e)
for extra in ("pmc_insts_zstd", "pmc_insts_encode"):
    if os.path.exists(os.path.join(G, f"{tag}_{extra}.txt")):
        open(os.path.join(P, f"{tag}_{extra}.txt"), "w").write(open(os.path.join(G, f"{tag}_{extra}.txt")).read())
if os.path.exists(os.path.join(G, f"{tag}_bench.json")):
    open(os.path.join(P, f"{tag}_bench_rocprof_run.json"), "w").write(open(os.path.join(G, f"{tag}_bench.json")).read())
if os.path.exists(os.path.join(G, f"{tag}_mixed_131072.json")):
    open(os.path.join(P, f"{tag}_bench_mixed_131072_units.json"), "w").write(open(os.path.join(G, f"{tag}_mixed_131072.json")).read())
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items() if isinstance(v, dict)}))
