"""Diagnostic: lz77_kernel's event counters (-DCHIP_STATS build).  COMPU_HIP_LIB=compu_amd/libcompu_hip_stats.so python tools/exp/x_stats.py [kind] [units]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("COMPU_HIP_LIB", os.path.join(ROOT, "compu_amd", "libcompu_hip_stats.so"))
import torch
import compu_amd
from bench_support import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda:0")
pay = synth.payloads(n)
packed, offs, lens = synth.deflate_units(pay, n, kind=kind)
stats = torch.zeros(n * 24, dtype=torch.int64, device=dev)
os.environ["CHIP_STATS_PTR"] = str(stats.data_ptr())
d_out = torch.zeros(n * 65536, dtype=torch.uint8, device=dev)
args = (-15, torch.from_numpy(packed).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
        d_out, torch.arange(n, dtype=torch.int64, device=dev) * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
compu_amd.decode_batch(*args)
torch.cuda.synchronize()
stats.zero_()
ol, iu, st = compu_amd.decode_batch(*args)
torch.cuda.synchronize()
assert (st == 2).all() and torch.equal(d_out, torch.from_numpy(pay).to(dev))
s = stats.cpu().numpy().reshape(n, 24).astype(np.float64).mean(axis=0)
names = ["pool passes", "passes without a copy", "matches by lanes", "matches by the wave", "trips: place unknown", "cycles: waiting for tokens", "cycles: passes", "cycles: groups",
         "cycles: set-up", "cycles: asleep", "cycles: total (8 waves)", "entries looked at"]
for i, nm in enumerate(names):
    print(f"  {nm:28s} {s[i]:12.1f}")
