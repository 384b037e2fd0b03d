import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import compu_amd, zstd_ref
Z = zstd_ref.load()
alice = open("/root/repo/tests/golden/alice29.txt","rb").read()
n = 1000
data = alice[:n]; comp = zstd_ref.compress(Z, data, 3)
buf = np.zeros((len(comp)+7)&~3, np.uint8); buf[:len(comp)] = np.frombuffer(comp, np.uint8)
d_out = torch.zeros(n+64, dtype=torch.uint8, device="cuda:0")
ol, iu, st = compu_amd.decode_batch(100, torch.from_numpy(buf).cuda(), torch.tensor([0],dtype=torch.int64).cuda(), torch.tensor([len(comp)],dtype=torch.int32).cuda(), d_out, torch.tensor([0],dtype=torch.int64).cuda(), torch.tensor([n],dtype=torch.int32).cuda())
torch.cuda.synchronize()
print(int(st[0]), int(ol[0]))
