"""Timing of the zstd kernel alone on synthetic 64 KiB frames: python tools/time_zstd.py [units]
With ZSTATS=1 and a -DCHIP_STATS build (COMPU_HIP_LIB=compu_amd/libcompu_hip_stats.so) it also prints the kernel's own cycle counters
per frame (phases and trip counts; zstd.hip, ZT_* / ZC)."""
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
zstats = None
if os.environ.get("ZSTATS") == "1":
    zstats = torch.zeros(n * 24, dtype=torch.int64, device=dev)
    os.environ["CHIP_STATS_PTR"] = str(zstats.data_ptr())
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
if zstats is not None:
    z = zstats.cpu().numpy().reshape(n, 24).astype(np.float64).mean(axis=0)
    names = {0: "cycles: whole frame", 1: "cycles: Huffman literals (walk, path, storing pass)", 2: "cycles:   of it the storing pass", 3: "cycles: FSE state chain",
             4: "cycles: sequence chunks altogether (chain + parallel part + placement + execution)", 5: "cycles: execution phase A (literal bytes)",
             6: "cycles: execution phase B (matches)", 7: "cycles: XXH64", 8: "trips: chunks of 64 sequences (chain runs)", 9: "trips: phase A steps",
             10: "trips: phase B steps", 11: "trips: literal rounds", 12: "trips: literal walk steps",
             13: "cycles: Huffman table (weights, their FSE table, code table)", 14: "cycles: the three sequence tables (descriptions, builds, re-coding)",
             15: "cycles:   Huffman: description and FSE table of the weights", 16: "cycles:   Huffman: the weights' two-state decode", 17: "cycles:   Huffman: code table from the weights",
             18: "trips: literal rounds decoded a second time (no room for the rows, or more than 32 trips)", 19: "trips: literal rounds (again)"}
    for i in sorted(names):
        print(f"  {names[i]:85s} {z[i]:12.1f}" + (f"  ({100 * z[i] / z[0]:5.1f} %)" if i and (i < 8 or i >= 13) and z[0] else ""))
    print(f"  {'cycles: parallel part + offsets + placement (4 - 3 - 5 - 6)':85s} {z[4] - z[3] - z[5] - z[6]:12.1f}  ({100 * (z[4] - z[3] - z[5] - z[6]) / z[0]:5.1f} %)")
