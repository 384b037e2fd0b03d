"""One-off robustness campaign on a GPU box: many random DEFLATE streams (all block types, strategies, sizes,
truncations, bit flips, short outputs) through chip_decode_batch, checked against the oracle; and zstd frames
against the oracle.  Usage: python tools/fuzz_gpu.py [rounds] [seed]"""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import compu_amd
from oracle import oracle as O
from test_inflate_gpu import run_batch, oracle_batch
import zstd_ref

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
alice = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()


def mk(kind, n):
    if kind == 0: return rnd.randbytes(n)
    if kind == 1:
        s = rnd.randrange(0, max(1, len(alice) - n)); return alice[s:s + n]
    if kind == 2: return bytes(rnd.choice(b"ab") for _ in range(n))
    if kind == 3: return b"\0" * n
    if kind == 4: return bytes(min(255, int(rnd.expovariate(0.05))) for _ in range(n))
    if kind == 5:  # short periods and nested copies: many same-step source dependencies
        out = bytearray(rnd.randbytes(rnd.randrange(1, 8)))
        while len(out) < n:
            d = rnd.randrange(1, min(len(out), rnd.choice([3, 8, 40, 300])) + 1); l = rnd.randrange(3, 60)
            for _ in range(l): out.append(out[-d])
            if rnd.random() < 0.2: out += rnd.randbytes(rnd.randrange(1, 4))
        return bytes(out[:n])
    w = [rnd.randbytes(rnd.randrange(2, 12)) for _ in range(rnd.randrange(2, 40))]  # few words: skewed codes, long repeats
    out = bytearray()
    while len(out) < n: out += rnd.choice(w)
    return bytes(out[:n])


bad = 0
total = 0
for it in range(rounds):
    parts, caps = [], []
    for _ in range(600):
        n = rnd.choice([0, 1, 5, 60, 300, 2000, 9000, 33000, 65536, 65537, 100000, 200000])
        data = mk(rnd.randrange(7), n)
        wb = rnd.choice([-15, -15, 15, 31])
        co = zlib.compressobj(rnd.choice([0, 1, 3, 6, 9]), zlib.DEFLATED, wb, rnd.choice([1, 8, 9]), rnd.choice([0, 0, 1, 2, 3, 4]))
        comp = bytearray()
        pos = 0
        while pos < n:
            step = rnd.randrange(1, n + 1)
            comp += co.compress(data[pos:pos + step]); pos += step
            if rnd.random() < 0.2: comp += co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        comp += co.flush()
        mode = rnd.randrange(6)
        if mode == 0 and comp: comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        elif mode == 1: comp = comp[:rnd.randrange(len(comp) + 1)]
        cap = n if mode != 2 else rnd.randrange(0, n + 1)
        parts.append((wb, bytes(comp))); caps.append(cap)
    for wb in (-15, 15, 31):
        idx = [i for i, (w, _) in enumerate(parts) if w == wb]
        ps = [parts[i][1] for i in idx]; cs = [caps[i] for i in idx]
        outs, ol, iu, st = run_batch(torch, wb, ps, cs)
        ref = oracle_batch(wb, ps, cs)
        for j in range(len(ps)):
            total += 1
            r_out, r_used, r_st = ref[j]
            ok = outs[j] == r_out
            if st[j] == 1 and r_st == 0 and len(r_out) == cs[j]: pass
            elif len(ps[j]) == 0 and st[j] == 0 and r_st == 1: pass
            else:
                ok = ok and st[j] == r_st and (r_st != 2 or iu[j] == r_used)
            if not ok:
                bad += 1
                print("INFLATE MISMATCH", it, wb, j, len(ps[j]), cs[j], int(st[j]), r_st, len(outs[j]), len(r_out), flush=True)
    print(f"round {it}: {total} streams, {bad} mismatches", flush=True)
# zstd frames vs the oracle (comparison as in tests/test_zstd_gpu.py)
from test_zstd_gpu import oracle_zstd_batch
Z = zstd_ref.load()
zbad = 0
ztot = 0
for it in range(rounds):
    frames, caps = [], []
    for _ in range(300):
        n = rnd.choice([0, 10, 900, 5000, 40000, 65536, 131072, 300000])
        data = mk(rnd.randrange(7), n)
        comp = bytearray(zstd_ref.compress(Z, data, rnd.choice([1, 3, 9, 19]), rnd.random() < 0.7, rnd.random() < 0.8))
        mode = rnd.randrange(5)
        if mode == 0 and comp: comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        elif mode == 1: comp = comp[:rnd.randrange(len(comp) + 1)]
        frames.append(bytes(comp)); caps.append(n + 300)
    outs, ol, iu, st = run_batch(torch, 100, frames, caps, check_tail=False)
    ref = oracle_zstd_batch(frames, caps)
    for j in range(len(frames)):
        ztot += 1
        r_out, r_used, r_st = ref[j]
        if r_st == 1:
            ok = int(st[j]) in (1, -70, -20) and outs[j] == r_out[:len(outs[j])]
        else:
            ok = int(st[j]) == r_st and (r_st not in (0, 2) or outs[j] == r_out) and (r_st != 2 or iu[j] == r_used)
            if not ok and int(st[j]) == 1 and r_st < 0:
                # a block that is broken AND does not fit the output range: the kernel works in the caller's range and stops where the
                # range ends (NeedOutput, whole blocks handed on), libzstd and the oracle decode a block in a buffer of their own and find
                # the damage first (include/compu_hip.h, status of batch calls).  Accepted iff the frame is an error at any capacity.
                big = oracle_zstd_batch([frames[j]], [1 << 24])[0]
                ok = big[2] < 0 and outs[j] == big[0][:len(outs[j])]
        if not ok:
            zbad += 1
            print("ZSTD MISMATCH", it, j, len(frames[j]), caps[j], int(st[j]), r_st, len(outs[j]), len(r_out), int(iu[j]), r_used, flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)  # the frame itself, for a look on the CPU afterwards
            open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_zstd_{seed}_{it}_{j}_cap{caps[j]}.bin"), "wb").write(frames[j])
    print(f"zstd round {it}: {ztot} frames, {zbad} mismatches", flush=True)
print("DONE", total, bad, ztot, zbad)
