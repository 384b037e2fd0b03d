"""Diagnostic: run the -DCHIP_STATS build of the library on a synthetic batch and print where a unit's
cycles go.  Usage (GPU box): COMPU_HIP_LIB=compu_amd/libcompu_hip_stats.so python tools/stats_run.py [kind] [units]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("COMPU_HIP_LIB", os.path.join(ROOT, "compu_amd", "libcompu_hip_stats.so"))
import torch  # noqa: E402

import compu_amd  # noqa: E402
from bench_support import synth  # noqa: E402

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
for _ in range(2):
    ol, iu, st = compu_amd.decode_batch(*args)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
compu_amd.decode_batch(*args)
b.record()
torch.cuda.synchronize()
assert (st == 2).all() and torch.equal(d_out, torch.from_numpy(pay).to(dev))
s = stats.cpu().numpy().reshape(n, 24).astype(np.float64)
names = ["hdr other", "window load", "walk", "resolve", "hdr:codelens", "hdr:build", "trailer", "-", "super-rounds", "path pieces", "tokens", "walk trips", "copy passes",
         "chunks", "match rounds", "wave copies", "fl:token groups", "fl:match rounds", "fl:chunk store", "-", "fl:tail", "lanes walking", "-", "-"]
tot = s[:, :8].sum(axis=1).mean() + s[:, 16:21].sum(axis=1).mean()
print(f"kind={kind} units={n} kernel={a.elapsed_time(b):.3f} ms; mean cycles/unit {tot:.0f}")
for i, nm in enumerate(names):
    if nm == "-":
        continue
    m = s[:, i].mean()
    print(f"  {nm:14s} {m:12.1f}" + (f"  ({100 * m / tot:5.1f}%)" if (i < 8 or i >= 16) else ""))
print(f"  super-rounds/unit {s[:, 8].mean():.2f}; walk trips/unit {s[:, 11].mean():.0f} (x4 tokens); lanes on the path per super-round {s[:, 9].sum() / s[:, 8].sum():.1f}; tokens/unit {s[:, 10].mean():.0f}; passes per match round {s[:, 12].sum() / max(1, s[:, 14].sum()):.2f}")
