"""Debug aid: dynamic-level GPU encoder vs the oracle, reporting every mismatch."""
import os, random, sys, zlib
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import compu_amd as c
from oracle import oracle as O
from test_inflate_gpu import _mk
alice = open(os.path.join(os.path.dirname(__file__), "..", "..", "tests/golden/alice29.txt"), "rb").read()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 77)
dev = "cuda:0"
bad = 0
for fmt, strategy in ((-15, 0), (15, 0), (31, 0), (-15, 2), (15, 3), (31, 1)):
    datas, desc = [], []
    for it in range(90):
        n = rnd.choice([0, 1, 3, 4, 5, 63, 64, 65, 100, 1000, 5000, 16319, 16320, 16321, 16384, 40000, 65535, 65536, 70000, 200000])
        k = rnd.randrange(5)
        datas.append(_mk(k, n, rnd, alice)); desc.append((k, n))
    datas.append((alice * 6)[:900000]); desc.append(("alice6", 900000))
    datas.append(os.urandom(40000) + alice[:60000] + bytes(50000) + os.urandom(30000)); desc.append(("mix", 180000))
    level = rnd.randrange(2, 10)
    lens = np.array([len(d) for d in datas], np.int32)
    offs = np.zeros(len(datas), np.int64)
    offs[1:] = np.cumsum(((lens[:-1].astype(np.int64) + 3) & ~3) + rnd.choice([0, 1, 2, 3]))
    buf = np.zeros(int(offs[-1] + lens[-1]) + 8, np.uint8)
    for d, o in zip(datas, offs):
        buf[o : o + len(d)] = np.frombuffer(d, np.uint8)
    caps = np.array([c.encode_bound(fmt, len(d)) for d in datas], np.int32)
    ooff = np.zeros(len(datas), np.int64)
    ooff[1:] = np.cumsum(caps[:-1].astype(np.int64) + 7)
    d_out = torch.full((int(ooff[-1] + caps[-1]) + 8,), 0xA5, dtype=torch.uint8, device=dev)
    out_len, status = c.encode_batch(fmt, level, torch.from_numpy(buf[: (len(buf) // 4) * 4]).to(dev), torch.from_numpy(offs).to(dev),
                                     torch.from_numpy(lens).to(dev), d_out, torch.from_numpy(ooff).to(dev), torch.from_numpy(caps).to(dev), strategy=strategy)
    torch.cuda.synchronize()
    h = d_out.cpu().numpy()
    ol, st = out_len.cpu().numpy(), status.cpu().numpy()
    for i, d in enumerate(datas):
        comp = bytes(h[ooff[i] : ooff[i] + ol[i]])
        e = O.DeflateEncoder(fmt, level, strategy)
        ref, ir, orr, est = e.encode(d, int(caps[i]) + 64, O.OP_FINISH)
        if comp != ref or st[i] != 2:
            bad += 1
            k = next((j for j in range(min(len(comp), len(ref))) if comp[j] != ref[j]), min(len(comp), len(ref)))
            hdr = {-15: 0, 15: 2, 31: 10}[fmt]
            print("MISMATCH", fmt, strategy, level, i, desc[i], "st", st[i], "len", len(comp), len(ref), "cap", caps[i], "first diff", k,
                  "btype gpu/ref", (comp[hdr] >> 1) & 3 if len(comp) > hdr else None, (ref[hdr] >> 1) & 3, flush=True)
print("bad", bad)
