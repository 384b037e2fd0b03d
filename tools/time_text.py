"""Timing on text-like units (64 KiB windows of tests/golden/alice29.txt, zlib level 6): python tools/time_text.py [units]"""
import os, sys, zlib, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import compu_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
alice = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()
rnd = random.Random(1)
distinct = []
for _ in range(256):
    o = rnd.randrange(0, len(alice) - 65536)
    d = alice[o:o + 65536]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    distinct.append((d, co.compress(d) + co.flush()))
parts = [distinct[i % 256][1] for i in range(n)]
lens = np.array([len(p) for p in parts], np.int32)
offs = np.zeros(n, np.int64); offs[1:] = np.cumsum(lens[:-1].astype(np.int64))
tot = int(lens.astype(np.int64).sum())
buf = np.zeros((tot + 7) & ~3, np.uint8); buf[:tot] = np.frombuffer(b"".join(parts), np.uint8)
dev = "cuda:0"
d_out = torch.zeros(n * 65536, dtype=torch.uint8, device=dev)
args = (-15, torch.from_numpy(buf).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev), d_out,
        torch.arange(n, dtype=torch.int64, device=dev) * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
for _ in range(2): compu_amd.decode_batch(*args)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ol, iu, st = compu_amd.decode_batch(*args); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ok = bool((st == 2).all()) and bytes(d_out[:65536].cpu().numpy()) == distinct[0][0]
print(f"text units (ratio {tot / (n * 65536):.3f}): {n} units {min(ts):.3f} ms = {n * 65536 / min(ts) / 1e6:.1f} GB/s decompressed (correct={ok})")
