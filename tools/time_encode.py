"""Timing of the batch encoder (a variant build may be chosen with COMPU_HIP_LIB): python tools/time_encode.py [units] [level]
ESTATS=1 with a -DCHIP_STATS build (COMPU_HIP_LIB=compu_amd/libcompu_hip_stats.so) prints the level-1 kernel's own cycle counters per phase.
Prints the best of five launches, the ratio, and whether the PRODUCT inflater (the default build) decodes the streams to the input."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import compu_amd  # noqa: E402
from bench_support import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
cache = f"/tmp/time_encode_{n}.npy"
if os.path.exists(cache):
    pay = np.load(cache)
else:
    pay = synth.payloads(n)
    np.save(cache, pay)
cap = (compu_amd.encode_bound(-15, 65536) + 15) & ~15
d_in = torch.from_numpy(pay).to(dev)
d_out = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
ar = torch.arange(n, dtype=torch.int64, device=dev)
args = (-15, level, d_in, ar * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev), d_out, ar * cap, torch.full((n,), cap, dtype=torch.int32, device=dev))
estats = None
if os.environ.get("ESTATS") == "1":  # with a -DCHIP_STATS build: the level-1 kernel's own cycle counters
    estats = torch.zeros(n * 24, dtype=torch.int64, device=dev)
    os.environ["CHIP_STATS_PTR"] = str(estats.data_ptr())
for _ in range(2):
    compu_amd.encode_batch(*args)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ol, st = compu_amd.encode_batch(*args)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
if estats is not None:  # (read before the inflater of the same diagnostic build writes its own counters there)
    estats_host = estats.cpu().numpy().copy()
    os.environ.pop("CHIP_STATS_PTR")
back = torch.zeros(n * 65536, dtype=torch.uint8, device=dev)
dl, iu, ds = compu_amd.decode_batch(-15, d_out, ar * cap, ol.to(torch.int32), back, ar * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
ok = bool((st == 2).all()) and bool((ds == 2).all()) and torch.equal(back, d_in)
print(f"{os.path.basename(os.environ.get('COMPU_HIP_LIB', 'prod'))}: encode level {level}, {n} units: {min(ts):.3f} ms, ratio {int(ol.to(torch.int64).sum()) / (n * 65536):.4f} (round trip={ok})")
if estats is not None:
    z = estats_host.reshape(n, 24).astype(np.float64).mean(axis=0)
    names = {0: "cycles: whole unit", 1: "cycles: look-up of the next chunk (hash, table, loads issued, table update rounds)", 2: "cycles: wait for the candidates' bytes, measure",
             3: "cycles: greedy choice (scalar walk over the chosen matches)", 4: "cycles: codes, prefix sum, bits into the LDS buffer", 5: "cycles: flush of the bit buffer",
             8: "trips: chunks", 9: "count: match candidates of at least four bytes", 10: "count: tokens emitted"}
    for i in sorted(names):
        print(f"  {names[i]:90s} {z[i]:12.1f}" + (f"  ({100 * z[i] / z[0]:5.1f} %)" if 0 < i < 8 and z[0] else ""))
