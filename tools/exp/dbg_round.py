import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("COMPU_HIP_LIB", os.path.join(ROOT, "compu_amd", "libcompu_hip_stats.so"))
import torch, compu_amd
from bench_support import synth
n = 2; unit = 2048
dev = torch.device("cuda:0")
pay = synth.payloads(n, unit_size=unit)
packed, offs, lens = synth.deflate_units(pay, n, unit_size=unit, kind="dynamic")
stats = torch.zeros(n * 24, dtype=torch.int64, device=dev)
os.environ["CHIP_STATS_PTR"] = str(stats.data_ptr())
d_out = torch.zeros(n * unit, dtype=torch.uint8, device=dev)
args = (-15, torch.from_numpy(packed).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
        d_out, torch.arange(n, dtype=torch.int64, device=dev) * unit, torch.full((n,), unit, dtype=torch.int32, device=dev))
compu_amd.decode_batch(*args)
torch.cuda.synchronize()
s = stats.cpu().numpy().reshape(n, 24).astype(np.uint64)
for u in range(n):
    for l, v in ((0, int(s[u, 22])), (1, int(s[u, 23]))):
        print(f"unit {u} lane {l}: nst {v >> 48} why {(v >> 40) & 255} p-B {(v >> 16) & 0xffffff} jl {(v >> 8) & 255} a_join {v & 255}")
    v = int(s[u, 19]); w = int(s[u, 6]); x = int(s[u, 0])
    print(f"   node1 {v >> 48} node2 {(v >> 40) & 255} cnt0 {(v >> 24) & 0xffff} cnt1 {(v >> 8) & 0xffff} B&31 {v & 255}; lane1: e_a0 {w >> 32} e_nst {w & 0xffffffff}; bm words of segment 1: {x >> 32:#x} {x & 0xffffffff:#x}")
