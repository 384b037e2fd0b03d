"""Which way the zstd literal decoder takes (diagnostic build: COMPU_HIP_LIB=compu_amd/libcompu_hip_stats.so): rounds copied out of the walk's rows
against rounds decoded a second time, on the data kinds of tests/test_zstd_gpu.py::test_literal_paths_rows_and_second_decode."""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import compu_amd
import zstd_ref
z = zstd_ref.load()
rnd = random.Random(21)
dev = torch.device("cuda:0")
for kind, name in enumerate(["one dominant byte", "two dominant bytes", "flat over 64 symbols", "3000-byte frames, flat"]):
    n = 65536 if kind < 3 else 3000
    parts = []
    for it in range(64):
        if kind == 0: data = bytes(97 if rnd.random() < 0.88 else rnd.randrange(256) for _ in range(n))
        elif kind == 1: data = bytes(rnd.choice(b"ab") if rnd.random() < 0.9 else rnd.randrange(256) for _ in range(n))
        else: data = bytes(32 + rnd.randrange(64) for _ in range(n))
        parts.append(zstd_ref.compress(z, data, 1, True, True))
    lens = np.array([len(p) for p in parts], np.int32)
    offs = np.zeros(64, np.int64); offs[1:] = np.cumsum(((lens[:-1] + 3) & ~3).astype(np.int64))
    buf = np.zeros(int(offs[-1]) + ((int(lens[-1]) + 7) & ~3), np.uint8)
    for i, p in enumerate(parts): buf[offs[i]:offs[i] + len(p)] = np.frombuffer(p, np.uint8)
    stats = torch.zeros(64 * 24, dtype=torch.int64, device=dev)
    os.environ["CHIP_STATS_PTR"] = str(stats.data_ptr())
    out = torch.zeros(64 * n, dtype=torch.uint8, device=dev)
    ol, iu, st = compu_amd.decode_batch(100, torch.from_numpy(buf).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev), out,
                                        torch.arange(64, dtype=torch.int64, device=dev) * n, torch.full((64,), n, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    s = stats.cpu().numpy().reshape(64, 24)
    print(f"{name:26s}: status ok {bool((st == 2).all())}; literal rounds {int(s[:, 19].sum())}, of them decoded a second time {int(s[:, 18].sum())}")
