"""PCIe-inclusive rate of the host-memory batch call (never bench.py's `value`): python tools/host_path_rate.py [units] [slice MiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import compu_amd
from bench_support import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
slice_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pay = synth.payloads(n)
packed, offs, lens = synth.deflate_units(pay, n, kind="dynamic")
h_in = torch.from_numpy(packed).pin_memory()
h_out = torch.empty(n * 65536, dtype=torch.uint8).pin_memory()
out_off = np.arange(n, dtype=np.uint64) * 65536
cap = np.full(n, 65536, np.uint32)
best = None
for it in range(4):
    t0 = time.perf_counter()
    ol, iu, st = compu_amd.decode_batch_host(-15, h_in.numpy(), offs.astype(np.uint64), lens.astype(np.uint32), h_out.numpy(), out_off, cap, slice_bytes=slice_mb << 20)
    dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
assert (st == 2).all() and np.array_equal(h_out.numpy(), pay)
tot_in = int(lens.sum())
print(f"host path, {n} units, slices of {slice_mb} MiB: {best * 1e3:.1f} ms = {n * 65536 / best / 1e9:.1f} GB/s decompressed "
      f"(H2D {tot_in / 1e9:.2f} GB + D2H {n * 65536 / 1e9:.2f} GB over the link: {(tot_in + n * 65536) / best / 1e9:.1f} GB/s)")
