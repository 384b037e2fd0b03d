"""Small fixed workload for rocprofv3 runs: python tools/prof_run.py [kind] [units] [iters]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import compu_amd  # noqa: E402
from bench_support import synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
pay = synth.payloads(n)
if kind.startswith("encode"):  # encode, encode6, ...: the encoder at that level (default 1) over the raw payload
    level = int(kind[6:] or 1)
    cap = (compu_amd.encode_bound(-15, 65536) + 15) & ~15
    d_in = torch.from_numpy(pay).to(dev)
    d_out = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    ar = torch.arange(n, dtype=torch.int64, device=dev)
    for _ in range(iters):
        ol, st = compu_amd.encode_batch(-15, level, d_in, ar * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev), d_out, ar * cap,
                                        torch.full((n,), cap, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    assert (st == 2).all()
    print("ok", kind, n, "compressed bytes", int(ol.to(torch.int64).sum()))
    sys.exit(0)
packed, offs, lens = synth.deflate_units(pay, n, kind=kind)
d_out = torch.zeros(n * 65536, dtype=torch.uint8, device=dev)
args = (-15, torch.from_numpy(packed).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
        d_out, torch.arange(n, dtype=torch.int64, device=dev) * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
for _ in range(iters):
    ol, iu, st = compu_amd.decode_batch(*args)
torch.cuda.synchronize()
assert (st == 2).all() and torch.equal(d_out, torch.from_numpy(pay).to(dev))
print("ok", kind, n, "compressed bytes", int(lens.sum()))
