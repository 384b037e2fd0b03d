"""Randomized check of the streaming decoder (chip_decode: buffering, resume at block boundaries, output growth) on a GPU
box: random streams fed in random pieces into random output sizes must add up to what the oracle produces from the
whole stream in one call, with the same terminal status.  Usage: python tools/fuzz_stream.py [streams] [seed]"""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import compu_amd as compu
from oracle import oracle as O

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
alice = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()


def mk(n):
    out = bytearray()
    while len(out) < n:
        k = rnd.randrange(5)
        if k == 0: out += alice[rnd.randrange(len(alice) // 2):][:rnd.randrange(1, 60000)]
        elif k == 1: out += rnd.randbytes(rnd.randrange(1, 20000))
        elif k == 2: out += bytes([rnd.randrange(256)]) * rnd.randrange(1, 40000)
        elif k == 3: out += bytes(rnd.choice(b"ab") for _ in range(rnd.randrange(1, 5000)))
        else: out += (out[-rnd.randrange(1, min(len(out), 32768) + 1):][:rnd.randrange(1, 2000)] if out else b"x")
    return bytes(out[:n])


modes = {-15: (compu.ZlibMode.Deflate, O.MODE_DEFLATE), 15: (compu.ZlibMode.Zlib, O.MODE_ZLIB), 31: (compu.ZlibMode.Gzip, O.MODE_GZIP)}
bad = 0
for it in range(n_streams):
    n = rnd.choice([0, 7, 3000, 70000, 400000, 1500000])
    data = mk(n)
    wb = rnd.choice([-15, 15, 31])
    co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, wb, 8, rnd.choice([0, 0, 3, 4]))
    comp = bytearray()
    pos = 0
    while pos < n:
        step = rnd.randrange(1, n + 1)
        comp += co.compress(data[pos:pos + step]); pos += step
        if rnd.random() < 0.3: comp += co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
    comp += co.flush()
    kind = rnd.randrange(5)
    if kind == 0 and comp: comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
    elif kind == 1: comp = comp[:rnd.randrange(len(comp) + 1)]
    comp = bytes(comp)
    ref_out, ref_rem, _, ref_st, ref_err = O.InflateDecoder(modes[wb][1]).decode(comp, n + 70000)
    dec = compu.decoder_interface.zlib_hip(modes[wb][0])
    out = bytearray()
    pos = 0
    status = None
    # piece and output sizes scaled so that a stream takes at most a few thousand calls
    piece = max(rnd.choice([1, 100, 5000, 70000, 1 << 20]), len(comp) // 1500 + 1)
    obuf = bytearray(max(rnd.choice([1, 50, 4096, 100000, 1 << 20]), n // 1500 + 1))
    guard = 0
    while True:
        guard += 1
        if guard > 200000:
            status = "loop"
            break
        chunk = comp[pos:pos + rnd.randrange(1, piece + 1)]
        r = dec.decode(chunk, obuf)
        out += obuf[:len(obuf) - r.output_remain]
        if not r.is_ok():
            status = r.status.code if hasattr(r.status, "code") else r.status
            break
        pos += len(chunk) - r.input_remain
        if r.status == compu.DecodeStatus.Finished:
            status = 2
            break
        if pos >= len(comp) and r.output_remain == len(obuf) and r.status in (compu.DecodeStatus.NeedInput, compu.DecodeStatus.NeedOutput):
            # nothing left to feed and nothing came out: the stream is truncated (an empty call is zlib's Z_BUF_ERROR,
            # which compu reads as NeedOutput, src/decoder/mod.rs:481).  The oracle sees the whole input in ONE call: it
            # says NeedInput for a stream cut short -- and NeedOutput for an EMPTY stream, whose only call is that empty call.
            status = 1 if len(comp) == 0 else 0
            break
    want = ref_err if ref_err else ref_st
    got = status
    if hasattr(got, "__int__") and not isinstance(got, int): got = int(got)
    ok = bytes(out) == ref_out and (got == want or (isinstance(got, compu.DecodeError) and got == compu.DecodeError(want)))
    if not ok:
        bad += 1
        print("MISMATCH", it, wb, n, len(comp), kind, piece, len(obuf), repr(status), want, len(out), len(ref_out), flush=True)
    if it % 20 == 19: print(f"{it + 1} streams, {bad} mismatches", flush=True)
# ---- zstd frames through the streaming decoder
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zstd_ref
Z = zstd_ref.load()
zbad = 0
nz = max(1, n_streams // 3)
for it in range(nz):
    n = rnd.choice([0, 7, 3000, 70000, 400000, 1500000])
    data = mk(n)
    comp = bytearray(zstd_ref.compress(Z, data, rnd.choice([1, 3, 9]), rnd.random() < 0.7, rnd.random() < 0.8))
    kind = rnd.randrange(5)
    if kind == 0 and comp: comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
    elif kind == 1: comp = comp[:rnd.randrange(len(comp) + 1)]
    comp = bytes(comp)
    ref_out, ref_rem, _, ref_st, ref_err = O.ZstdDecoder().decode(comp, n + 70000)
    dec = compu.decoder_interface.zstd_hip()
    out = bytearray()
    pos = 0
    status = None
    piece = max(rnd.choice([1, 100, 5000, 70000, 1 << 20]), len(comp) // 1500 + 1)
    obuf = bytearray(max(rnd.choice([50, 4096, 100000, 1 << 20]), n // 1500 + 1))
    for guard in range(20000):
        chunk = comp[pos:pos + rnd.randrange(1, piece + 1)]
        r = dec.decode(chunk, obuf)
        out += obuf[:len(obuf) - r.output_remain]
        if not r.is_ok():
            status = r.status
            break
        pos += len(chunk) - r.input_remain
        if r.status == compu.DecodeStatus.Finished:
            status = 2
            break
        if pos >= len(comp) and r.output_remain == len(obuf):
            status = 0
            break
    want = ref_err if ref_err else ref_st
    if want == 2 or want == 0:
        ok = bytes(out) == ref_out and status == want
    else:  # corrupt: an error, with whatever was produced before it a prefix of the oracle's or vice versa
        ok = isinstance(status, compu.DecodeError) or status == 0
    if not ok:
        zbad += 1
        print("ZSTD MISMATCH", it, n, len(comp), kind, piece, len(obuf), repr(status), want, len(out), len(ref_out), flush=True)
print("DONE", n_streams, bad, nz, zbad)
