"""Pins the CPU oracle (oracle/oracle_zstd.c): the reference's zstd fixtures through the phases of
tests/decoder.rs:21-77 (should_decode_zstd :119-128), and the system libzstd on generated frames."""
import random

import pytest

import zstd_ref
from conftest import golden
from oracle import oracle as O

Z = zstd_ref.load()
needs_libzstd = pytest.mark.skipif(Z is None, reason="no system libzstd to cross-check against")


@pytest.mark.parametrize("name", ["10x10y", "alice29.txt"])
def test_reference_zstd_fixtures(name):
    data, comp = golden(name), golden(name + ".compressed.zstd")
    d = O.ZstdDecoder()
    assert d.decode(comp, len(data)) == (data, 0, 0, O.FINISHED, 0)  # tests/decoder.rs:25-31
    d.reset()
    out1, ir, orr, st, err = d.decode(comp, 1)  # :34-36
    assert (st, orr, err) == (O.NEED_OUTPUT, 0, 0)
    out2, ir, orr, st, err = d.decode(comp[len(comp) - ir :], len(data) - 1)  # :40-44
    assert (out1 + out2, st) == (data, O.FINISHED)
    d.reset()
    pos, acc = 0, b""
    while True:  # :47-63
        got, ir, orr, st, err = d.decode(comp[pos:], 4096)
        assert not err
        acc += got
        pos = len(comp) - ir
        if st == O.FINISHED:
            break
    assert acc == data
    assert O.lib().orc_zstd_strerror(0) == b"No error detected"  # :74-76
    # checksum and corruption classes of SURVEY.md sec. 8b
    bad = bytearray(comp)
    bad[-1] ^= 1
    assert O.ZstdDecoder().decode(bytes(bad), len(data) + 10)[4] == -22
    assert O.ZstdDecoder().decode(b"\x00\x01\x02\x03\x04", 10)[4] == -10


def test_xxh64_known_answers():
    # published XXH64 test vectors (seed 0)
    assert O.xxh64(b"") == 0xEF46DB3751D8E999
    assert O.xxh64(b"a") == 0xD24EC4F1A98C6E5B
    assert O.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert O.xxh64(b"Nobody inspects the spammish repetition") == 0xFBCEA83C8A378BF1


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


@needs_libzstd
def test_generated_frames_roundtrip_chunked(alice):
    rnd = random.Random(4)
    for it in range(150):
        n = rnd.choice([0, 1, 2, 10, 100, 1000, 5000, 65536, 70000, 150000, 300000])
        data = _mk(rnd.randrange(5), n, rnd, alice)
        comp = zstd_ref.compress(Z, data, rnd.choice([1, 3, 5, 9, 15, 19, -1]), rnd.random() < 0.7, rnd.random() < 0.8)
        d, pos, out = O.ZstdDecoder(), 0, b""
        chunk, ocap = rnd.choice([3, 100, 4096, 10**9]), rnd.choice([100, 4096, 10**6])
        stream = comp + b"zz"
        while True:
            piece = stream[pos : pos + chunk]
            got, ir, orr, st, err = d.decode(piece, ocap)
            assert not err
            out += got
            pos += len(piece) - ir
            if st == O.FINISHED:
                break
        assert out == data and pos == len(comp)


@needs_libzstd
def test_truncated_and_corrupt_frames_vs_libzstd(alice):
    """Truncations must agree exactly with ZSTD_decompressStream (incl. raw blocks streaming through).
    Bit flips must agree except where the system libzstd (1.4.x) is laxer than RFC 8878: it clamps an
    over-read of the last Huffman symbol and does not insist that bitstreams end exactly, so it may
    reach the checksum (or finish) where the oracle already reports corruption; and the order in
    which two simultaneous faults are reported may differ."""
    rnd = random.Random(6)
    lenient = 0
    for it in range(1200):
        n = rnd.choice([50, 500, 5000, 70000])
        data = _mk(rnd.choice([1, 2, 4, 0]), n, rnd, alice)
        comp = bytearray(zstd_ref.compress(Z, data, rnd.choice([1, 3, 9]), rnd.random() < 0.7, rnd.random() < 0.8))
        mode = rnd.randrange(3)
        if mode == 0:
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            comp = comp[: rnd.randrange(len(comp))]
        zo, zir, zor, zst, zerr = zstd_ref.stream_decode_once(Z, bytes(comp), n + 100)
        out, ir, orr, st, err = O.ZstdDecoder().decode(bytes(comp), n + 100)
        if mode != 0:
            assert (out, st, err) == (zo, zst, zerr), (it, mode)
            continue
        if (st, err) == (zst, zerr):
            assert st is None or out == zo
        else:
            assert err in (-20, -70) or zerr in (-20, -70), (it, st, err, zst, zerr)
            lenient += 1
    assert lenient < 30
