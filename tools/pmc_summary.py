"""Sum rocprofv3 PMC counters per kernel: python tools/pmc_summary.py <dir>/<prefix>_counter_collection.csv [units]"""
import csv, sys, collections
path = sys.argv[1]
units = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not any(t in k for t in ("inflate", "zstd", "deflate", "tokens_kernel", "lz77")):
        continue
    print(k[:60])
    for c, v in sorted(cs.items()):
        last = v[-1]
        print(f"   {c:24s} last dispatch {last:16.0f}   per unit {last / units:12.1f}")
