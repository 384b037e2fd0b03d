"""Timing of the zstd kernel alone on synthetic 64 KiB frames: python tools/time_zstd.py [units]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import compu_amd
from bench_support import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
pay = synth.payloads(n)
mv = memoryview(pay)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    parts = list(ex.map(synth._mixed_one, [(mv[i * 65536:(i + 1) * 65536], 2 * i + 0) for i in range(n)]))
# force zstd for every unit: pick indices whose splitmix parity is even
parts = []
i = 0
idx = []
while len(idx) < n:
    if not (synth._splitmix64(i) & 1):
        idx.append(i)
    i += 1
with ThreadPoolExecutor(16) as ex:
    parts = list(ex.map(synth._mixed_one, [(mv[k * 65536:(k + 1) * 65536], idx[k]) for k in range(n)]))
lens = np.array([len(p) for p in parts], np.int32)
offs = np.zeros(n, np.int64); offs[1:] = np.cumsum(lens[:-1].astype(np.int64))
buf = np.zeros((int(lens.astype(np.int64).sum()) + 7) & ~3, np.uint8); buf[: int(lens.astype(np.int64).sum())] = np.frombuffer(b"".join(parts), np.uint8)
d_out = torch.zeros(n * 65536, dtype=torch.uint8, device=dev)
args = (100, torch.from_numpy(buf).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev), d_out,
        torch.arange(n, dtype=torch.int64, device=dev) * 65536, torch.full((n,), 65536, dtype=torch.int32, device=dev))
for _ in range(2):
    compu_amd.decode_batch(*args)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ol, iu, st = compu_amd.decode_batch(*args); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ok = bool((st == 2).all()) and torch.equal(d_out, torch.from_numpy(pay).to(dev))
print(f"{os.path.basename(os.environ.get('COMPU_HIP_LIB','prod'))}: zstd {n} frames, ratio {lens.sum()/(n*65536):.3f}: {min(ts):.3f} ms (correct={ok})")
