"""Timing-only run of a (possibly deliberately broken, experiment) build: COMPU_HIP_LIB=... python tools/time_run.py [kind] [units]
CAUTION (round 4): the default 16 384 units are exactly four units per wave of a 16-waves-per-CU persistent grid; a variant that changes the
number of resident waves must be compared at the full launch size (tools/exp/w18_full.sh: bench.py --workload W at 65 536 units), or it
is judged by its ragged last round."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import compu_amd
from bench_support import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda:0")
cache = f"/tmp/time_run_{kind}_{n}.npz"  # variants timed back to back share one generated batch
if os.path.exists(cache):
    z = np.load(cache)
    pay, packed, offs, lens = z["pay"], z["packed"], z["offs"], z["lens"]
else:
    pay = synth.payloads(n)
    packed, offs, lens = synth.deflate_units(pay, n, kind=kind)
    np.savez(cache, pay=pay, packed=packed, offs=offs, lens=lens)
d_out = torch.zeros(n * 65536, dtype=torch.uint8, device=dev)
args = (-15, torch.from_numpy(packed).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
        d_out, torch.arange(n, dtype=torch.int64, device=dev) * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
for _ in range(2):
    compu_amd.decode_batch(*args)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ol, iu, st = compu_amd.decode_batch(*args); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ok = bool((st == 2).all()) and torch.equal(d_out, torch.from_numpy(pay).to(dev))
print(f"{os.path.basename(os.environ.get('COMPU_HIP_LIB','prod'))}: {kind} {n} units: {min(ts):.3f} ms (correct={ok})")
