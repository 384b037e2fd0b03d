"""Debug helper (GPU box): decode a few synthetic units and say where the output first differs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import compu_amd
from bench_support import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
unit = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = torch.device("cuda:0")
pay = synth.payloads(n, unit_size=unit)
packed, offs, lens = synth.deflate_units(pay, n, unit_size=unit, kind=kind)
d_out = torch.zeros(n * unit, dtype=torch.uint8, device=dev)
ol, iu, st = compu_amd.decode_batch(-15, torch.from_numpy(packed).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
                                    d_out, torch.arange(n, dtype=torch.int64, device=dev) * unit, torch.full((n,), unit, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
out = d_out.cpu().numpy().reshape(n, unit)
exp = pay.reshape(n, unit)
for i in range(n):
    neq = np.nonzero(out[i] != exp[i])[0]
    print(f"unit {i}: status {int(st[i])} out_len {int(ol[i])} in_used {int(iu[i])}/{int(lens[i])} first diff {int(neq[0]) if len(neq) else -1} ndiff {len(neq)}")
    if len(neq):
        k = int(neq[0]); print("   got", out[i][k:k+16].tolist(), "\n   exp", exp[i][k:k+16].tolist())
