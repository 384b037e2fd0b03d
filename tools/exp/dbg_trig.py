import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("COMPU_HIP_LIB", os.path.join(ROOT, "compu_amd", "libcompu_hip_stats.so"))
import torch, compu_amd
from bench_support import synth
n = 8
dev = torch.device("cuda:0")
pay = synth.payloads(n)
packed, offs, lens = synth.deflate_units(pay, n, kind="dynamic")
stats = torch.zeros(n * 24, dtype=torch.int64, device=dev)
os.environ["CHIP_STATS_PTR"] = str(stats.data_ptr())
d_out = torch.zeros(n * 65536, dtype=torch.uint8, device=dev)
args = (-15, torch.from_numpy(packed).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
        d_out, torch.arange(n, dtype=torch.int64, device=dev) * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
compu_amd.decode_batch(*args)
torch.cuda.synchronize()
s = stats.cpu().numpy().reshape(n, 24).astype(np.uint64)
for u in range(n):
    a, b, c = int(s[u, 13]), int(s[u, 14]), int(s[u, 15])
    print(f"unit {u}: rounds {int(s[u,8])} trig-ends {int(s[u,22])}; 1st trigger: seg {a >> 44} jseg {(a >> 24) & 0xfffff} stop {b >> 56} next_seg {(b >> 40) & 0xffff} nst {(b >> 24) & 0xffff} p-gbit {c >> 24} (seg of p {(c >> 24) >> 8}) jbits0 {int(s[u,12]) >> 32:#x} ebits0 {int(s[u,9]) >> 32:#x} G {int(s[u,19]) >> 32}")
