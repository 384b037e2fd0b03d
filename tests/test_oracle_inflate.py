"""Pins the CPU oracle (oracle/oracle_inflate.c): reference fixtures through the phases of
tests/decoder.rs:21-77, and system zlib (Python's zlib module) on generated, chunked, truncated and
corrupted streams -- outputs, input_remain, status class, error code and zlib's message."""
import os
import random
import re
import zlib

import pytest

from conftest import golden
from oracle import oracle as O


def _mk(kind, n, rnd, alice):
    if kind == 0:
        return rnd.randbytes(n)
    if kind == 1:
        s = rnd.randrange(0, max(1, len(alice) - n))
        return alice[s : s + n]
    if kind == 2:
        return bytes(rnd.choice(b"ab") for _ in range(n))
    if kind == 3:
        return b"\0" * n
    return bytes(min(255, int(rnd.expovariate(0.05))) for _ in range(n))


@pytest.mark.parametrize("name,ir_after_one_byte", [("10x10y", 13), ("alice29.txt", 54086)])
def test_reference_gzip_fixtures(name, ir_after_one_byte):
    data, comp = golden(name), golden(name + ".compressed.gz")
    for mode in (O.MODE_GZIP, O.MODE_AUTO):
        d = O.InflateDecoder(mode)
        out, ir, orr, st, err = d.decode(comp, len(data))  # tests/decoder.rs:25-31
        assert (out, ir, orr, st, err) == (data, 0, 0, O.FINISHED, 0)
        d.reset()
        out1, ir, orr, st, err = d.decode(comp, 1)  # :34-36 (DATA.len() / 2 == 1)
        assert (st, orr, err) == (O.NEED_OUTPUT, 0, 0)
        assert ir == ir_after_one_byte  # what stock zlib 1.2.11 leaves
        out2, ir, orr, st, err = d.decode(comp[len(comp) - ir :], len(data) - 1)  # :40-44
        assert (out1 + out2, st) == (data, O.FINISHED)
        d.reset()
        pos, acc = 0, b""
        while True:  # :47-63, Buffer::<4096>
            got, ir, orr, st, err = d.decode(comp[pos:], 4096)
            assert not err
            acc += got
            pos = len(comp) - ir
            if st == O.FINISHED:
                break
        assert acc == data
    assert O.lib().orc_zlib_strerror(0) is not None  # :74-76
    assert O.lib().orc_zlib_strerror(-3) == b"data error"


def test_wrong_wrapper_modes():
    comp = golden("10x10y.compressed.gz")
    for mode in (O.MODE_ZLIB, O.MODE_DEFLATE):
        _, _, _, st, err = O.InflateDecoder(mode).decode(comp, 100)
        assert err == -3
    z = zlib.compress(b"hello hello hello")
    assert O.InflateDecoder(O.MODE_GZIP).decode(z, 100)[4] == -3
    assert O.InflateDecoder(O.MODE_AUTO).decode(z, 100)[0] == b"hello hello hello"


def test_chunked_streams_match_zlib(alice):
    rnd = random.Random(2)
    for it in range(250):
        n = rnd.choice([0, 1, 2, 10, 100, 1000, 5000, 65536, 70000, 200000]) if it % 7 else rnd.randrange(0, 3000)
        data = _mk(rnd.randrange(5), n, rnd, alice)
        level, strat, wb = rnd.choice(range(10)), rnd.choice([0, 0, 0, 1, 2, 3, 4]), rnd.choice([-15, 15, 31])
        co = zlib.compressobj(level, zlib.DEFLATED, wb, rnd.choice([1, 8, 9]), strat)
        comp = co.compress(data[: n // 2]) + (co.flush(zlib.Z_SYNC_FLUSH) if rnd.random() < 0.3 else b"") + co.compress(data[n // 2 :]) + co.flush()
        mode = {-15: O.MODE_DEFLATE, 15: rnd.choice([O.MODE_ZLIB, O.MODE_AUTO]), 31: rnd.choice([O.MODE_GZIP, O.MODE_AUTO])}[wb]
        d = O.InflateDecoder(mode)
        z = zlib.decompressobj(47 if mode == O.MODE_AUTO else mode)
        trailing = rnd.randbytes(rnd.choice([0, 0, 3, 17]))
        chunk, ocap = rnd.choice([1, 2, 3, 7, 64, 1000, 10**9]), rnd.choice([1, 5, 100, 4096, 10**6])
        if n > 20000 and chunk < 7 and ocap < 100:
            ocap = 4096  # keep the CPU suite quick
        stream, pos, out = comp + trailing, 0, b""
        while True:
            piece = stream[pos : pos + chunk]
            got, ir, orr, st, err = d.decode(piece, ocap)
            zo = z.decompress(piece, ocap)
            assert got == zo and not err
            out += got
            pos += len(piece) - ir
            if st == O.FINISHED:
                assert z.eof and len(stream) - pos == len(trailing)
                break
            assert not z.eof and ir == len(z.unconsumed_tail)  # same input_remain as zlib
            assert piece or got
        assert out == data


def test_corrupt_streams_match_zlib_verdicts(alice):
    rnd = random.Random(3)
    for it in range(1500):
        n = rnd.choice([50, 500, 5000])
        data = _mk(rnd.choice([1, 2, 4]), n, rnd, alice)
        wb = rnd.choice([-15, 15, 31])
        co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, wb, 8, rnd.choice([0, 4]))
        comp = bytearray(co.compress(data) + co.flush())
        k = rnd.randrange(len(comp))
        comp[k] ^= 1 << rnd.randrange(8)
        if rnd.random() < 0.2:
            comp = comp[: rnd.randrange(len(comp))]
        comp = bytes(comp)
        try:
            z = zlib.decompressobj(wb)
            zo, zerr, zeof = z.decompress(comp, 10**6), None, z.eof
        except zlib.error as e:
            zo, zerr, zeof = None, str(e), False
        d = O.InflateDecoder(wb)
        got, ir, orr, st, err = d.decode(comp, 10**6)
        if zerr:
            m = re.match(r"Error (-?\d+) while decompressing data: (.*)", zerr)
            assert err == int(m.group(1)) and d.msg() == m.group(2), (it, zerr, err, d.msg())
        else:
            assert not err and got == zo and (st == O.FINISHED) == zeof


def test_checksums():
    rnd = random.Random(9)
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 5551, 5552, 5553, 100000):
        b = rnd.randbytes(n)
        assert O.crc32(b) == zlib.crc32(b)
        assert O.adler32(b) == zlib.adler32(b)


def test_units_helper_matches_streams(alice):
    import numpy as np

    rnd = random.Random(1)
    parts = [zlib.compress(_mk(1, 3000, rnd, alice), 6, -15) for _ in range(40)] if hasattr(zlib, "x") else None
    parts = []
    datas = []
    for _ in range(40):
        dat = _mk(rnd.randrange(5), rnd.choice([0, 10, 3000, 65536]), rnd, alice)
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        parts.append(co.compress(dat) + co.flush())
        datas.append(dat)
    lens = np.array([len(p) for p in parts], np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    buf = np.frombuffer(b"".join(parts), np.uint8)
    caps = np.array([len(d) for d in datas], np.uint32)
    ooff = np.concatenate([[0], np.cumsum(caps[:-1], dtype=np.uint64)]).astype(np.uint64)
    out, out_len, status, bad = O.inflate_units(O.MODE_DEFLATE, buf, offs, lens, int(caps.sum()), ooff, caps, threads=3)
    assert bad == 0 and (status == O.FINISHED).all() and (out_len == caps).all()
    assert out[: int(caps.sum())].tobytes() == b"".join(datas)
