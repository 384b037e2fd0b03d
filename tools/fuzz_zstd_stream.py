"""Randomized check of the bounded zstd streaming decoder (chip_decode with format zstd: block checkpoints, running XXH64,
input / output compaction, buffer growth) on a GPU box: windowed frames of a few to a few dozen MiB, random window, checksum,
content size, long-distance matching and level, fed in random pieces into random output sizes, must reproduce the data (CRC-32 and
length) and end Finished; damaged frames must end in an error or NeedInput, never in wrong data handed on as a finished frame.
Usage: python tools/fuzz_zstd_stream.py [frames] [seed]"""
import ctypes as C
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import compu_amd as compu
import zstd_ref

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
alice = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()
z = zstd_ref.load()


def mk(n):
    out = bytearray()
    while len(out) < n:
        k = rnd.randrange(6)
        if k == 0: out += alice[rnd.randrange(len(alice) // 2):][:rnd.randrange(1, 200000)]
        elif k == 1: out += rnd.randbytes(rnd.randrange(1, 300000))
        elif k == 2: out += bytes([rnd.randrange(256)]) * rnd.randrange(1, 400000)
        elif k == 3: out += bytes(rnd.choice(b"abc") for _ in range(rnd.randrange(1, 20000)))
        elif k == 4 and out:  # a far copy: up to 6 MiB back
            back = rnd.randrange(1, min(len(out), 6 << 20) + 1)
            out += out[-back:][:rnd.randrange(1, 500000)]
        else: out += rnd.randbytes(64) * rnd.randrange(1, 3000)
    return out[:n]


def compress(data, level, wlog, checksum, csize, ldm):
    cctx = z.ZSTD_createCCtx()
    for k, v in ((100, level), (101, wlog), (201, int(checksum)), (200, int(csize)), (160, int(ldm))):
        z.ZSTD_CCtx_setParameter(cctx, k, v)
    cap = z.ZSTD_compressBound(len(data))
    dst = (C.c_char * cap)()
    src = (C.c_char * len(data)).from_buffer(data)
    n = z.ZSTD_compress2(cctx, dst, cap, src, len(data))
    z.ZSTD_freeCCtx(cctx)
    assert not z.ZSTD_isError(n)
    return bytes(memoryview(dst)[:n])


bad = 0
for it in range(n_frames):
    n = rnd.choice([300_000, 3 << 20, 9 << 20, 20 << 20, 40 << 20])
    data = mk(n)
    wlog = rnd.choice([17, 18, 19, 20, 21, 22, 23])
    checksum, csize, ldm = rnd.random() < 0.7, rnd.random() < 0.5, rnd.random() < 0.4
    comp = compress(data, rnd.choice([1, 2, 3, 5]), wlog, checksum, csize, ldm)
    want_crc = zlib.crc32(data)
    kind = rnd.randrange(6)  # 0: a flipped bit, 1: truncated, else intact
    if kind == 0:
        b = bytearray(comp); b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8); comp = bytes(b)
    elif kind == 1:
        comp = comp[:rnd.randrange(len(comp))]
    dec = compu.decoder_interface.zstd_hip()
    piece = max(rnd.choice([1000, 20000, 70000, 1 << 20, 4 << 20]), len(comp) // 3000 + 1)
    obuf = bytearray(max(rnd.choice([4096, 100000, 1 << 20, 3 << 20]), n // 3000 + 1))
    crc, got, pos, status, calls, peak, status_last = 0, 0, 0, None, 0, 0, None
    while True:
        calls += 1
        if calls > 100000:
            status = "loop"
            break
        # (a caller that was told NeedOutput drains before it feeds more, as compu's own loop does)
        chunk = b"" if status_last == compu.DecodeStatus.NeedOutput else comp[pos:pos + rnd.randrange(1, piece + 1)]
        r = dec.decode(chunk, obuf)
        status_last = r.status if r.is_ok() else None
        if not r.is_ok():
            status = r.status
            break
        k = len(obuf) - r.output_remain
        crc = zlib.crc32(memoryview(obuf)[:k], crc)
        got += k
        pos += len(chunk) - r.input_remain
        if calls % 16 == 0:
            peak = max(peak, dec.footprint()[1])
        if r.status == compu.DecodeStatus.Finished:
            status = "finished"
            break
        if r.status == compu.DecodeStatus.NeedInput and pos >= len(comp):
            status = "needinput"  # every byte of a truncated frame is in, everything decodable has been handed on
            break
    ok = True
    if kind >= 2:
        ok = status == "finished" and got == n and crc == want_crc and pos == len(comp)
        # O(window): the device side holds the window, a block, the piece and slack -- not the frame
        if ok and n >= (20 << 20) and peak > (1 << wlog) * 3 + len(obuf) + piece * 2 + (12 << 20):
            ok = False
    elif status == "finished":  # damage that the frame's checks cannot see (no checksum, or a flip in a skipped field): data must be right
        ok = (not checksum or crc == want_crc) and got == n if checksum else True
    elif status == "loop":
        ok = False
    if not ok:
        bad += 1
        print(f"MISMATCH frame {it}: kind {kind} n {n} wlog {wlog} ck {checksum} cs {csize} ldm {ldm} comp {len(comp)} piece {piece} obuf {len(obuf)} -> {status} got {got} pos {pos} peak {peak}")
    elif it % 5 == 0:
        print(f"frame {it}: kind {kind} n {n >> 20} MiB wlog {wlog} comp {len(comp)} calls {calls} peak dev {peak >> 20} MiB -> {status}", flush=True)
print(f"done: {n_frames} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
