import sys, os, ctypes, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import compu_amd
import zstd_ref
Z = zstd_ref.load()
def zstd_compress(d, l): return zstd_ref.compress(Z, d, l)
rnd = random.Random(3)
alice = open("/root/repo/tests/golden/alice29.txt","rb").read()
for n in [100, 1000, 3000, 10000, 30000, 70000, 131072]:
    for lvl in (1, 3, 19):
        data = alice[:n]
        comp = zstd_compress(data, lvl)
        buf = np.zeros((len(comp)+7)&~3, np.uint8); buf[:len(comp)] = np.frombuffer(comp, np.uint8)
        d_out = torch.zeros(n+64, dtype=torch.uint8, device="cuda:0")
        ol, iu, st = compu_amd.decode_batch(100, torch.from_numpy(buf).cuda(), torch.tensor([0],dtype=torch.int64).cuda(), torch.tensor([len(comp)],dtype=torch.int32).cuda(), d_out, torch.tensor([0],dtype=torch.int64).cuda(), torch.tensor([n],dtype=torch.int32).cuda())
        ok = bytes(d_out[:n].cpu().numpy()) == data
        print(n, lvl, len(comp), int(st[0]), int(ol[0]), ok)
