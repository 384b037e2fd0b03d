"""Debug: which units / bytes of a synthetic batch differ through the two-kernel pipeline."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import compu_amd
from bench_support import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
pay = synth.payloads(n)
packed, offs, lens = synth.deflate_units(pay, n, kind=kind)
d_pay = torch.from_numpy(pay).to(dev)
args = lambda out: (-15, torch.from_numpy(packed).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
        out, torch.arange(n, dtype=torch.int64, device=dev) * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
for r in range(reps):
    d_out = torch.full((n * 65536,), 0xAA, dtype=torch.uint8, device=dev)
    ol, iu, st = compu_amd.decode_batch(*args(d_out))
    torch.cuda.synchronize()
    neq = (d_out != d_pay).view(n, 65536)
    badu = neq.any(dim=1).nonzero().flatten().cpu().numpy()
    print(f"rep {r}: bad units {len(badu)} of {n}; status != 2: {int((st != 2).sum())}; out_len != 65536: {int((ol != 65536).sum())}")
    for u in badu[:6]:
        idx = neq[u].nonzero().flatten().cpu().numpy()
        got = d_out.view(n, 65536)[u].cpu().numpy(); want = pay.reshape(n, 65536)[u]
        runs = np.split(idx, np.where(np.diff(idx) != 1)[0] + 1)
        print(f"  unit {u}: {len(idx)} bytes differ in {len(runs)} runs; first runs:", [(int(r[0]), len(r)) for r in runs[:8]])
        r0 = runs[0]
        print("    got ", got[r0[0]:r0[0] + 12].tolist(), "want", want[r0[0]:r0[0] + 12].tolist())
