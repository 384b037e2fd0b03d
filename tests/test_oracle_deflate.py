"""Pins the CPU oracle of the encoder (oracle/oracle_deflate.c), the exact statement of what the HIP encoder kernels
produce.  compu pins its encoder by round trip and cross-API determinism only (tests/encoder.rs:10-78, 115-173); here:
every level / strategy / mode decodes with system zlib to the input, the status contract of internal_zlib_impl_encode!
(src/encoder/mod.rs:334-370) on Process / Flush / Finish with small outputs, the block kinds, the ratio bounds that
DESIGN.md states, and the committed digests of tests/golden/encoder_digests.json (tools/make_encoder_digests.py)."""
import hashlib
import json
import os
import random
import zlib

import pytest

from conftest import GOLDEN, golden
from oracle import oracle as O
from test_oracle_inflate import _mk

WBITS = {O.MODE_DEFLATE: -15, O.MODE_ZLIB: 15, O.MODE_GZIP: 31}


def _enc(data, mode=O.MODE_DEFLATE, level=6, strategy=0):
    e = O.DeflateEncoder(mode, level, strategy)
    comp, ir, orr, st = e.encode(data, len(data) + len(data) // 500 + 4096, O.OP_FINISH)
    assert (ir, st) == (0, O.ENC_FINISHED)
    return comp


def test_committed_digests():
    want = json.load(open(os.path.join(GOLDEN, "encoder_digests.json")))
    assert len(want) == 14
    for key, w in want.items():
        name, mode, level, strategy = key.split(":")
        comp = _enc(golden(name), O.MODE_GZIP, int(level[5:]), int(strategy[8:]))
        assert (len(comp), hashlib.sha256(comp).hexdigest()) == (w["size"], w["sha256"]), key


def test_every_level_strategy_mode_round_trips(alice):
    rnd = random.Random(5)
    sizes = [0, 1, 3, 4, 5, 63, 64, 65, 127, 128, 129, 1000, 4096, 65535, 65536, 65537, 70000, 200000]
    for it in range(160):
        n = rnd.choice(sizes)
        data = _mk(rnd.randrange(5), n, rnd, alice)
        level, strategy, mode = rnd.randrange(0, 10), rnd.randrange(0, 5), rnd.choice(list(WBITS))
        comp = _enc(data, mode, level, strategy)
        assert zlib.decompress(comp, WBITS[mode]) == data, (it, n, level, strategy, mode)
        assert comp == _enc(data, mode, level, strategy)  # deterministic
        assert len(comp) <= n + 5 * (n // 65535 + 1) + 6 * (n // 65472 + 1) + 5 + 18  # chip_encode_bound


def test_sixteen_bit_table_positions_wrap_and_alias(alice):
    """The match finder keeps the low 16 bits of a position per hash slot (oracle_deflate.c, header): inputs far past 64 KiB, and
    inputs that repeat with a period of exactly / nearly 64 KiB (a slot then names the copy one period back, out of the window's
    reach, or a position that merely shares its low bits), still give valid streams -- system zlib decodes them to the input -- and
    find the matches that are there across the 64 KiB marks."""
    rnd = random.Random(9)
    for period in (65536, 65535, 65537, 70000, 32768, 32769, 131072):
        blk = (alice * 2)[: period // 2] + rnd.randbytes(period - period // 2)
        data = (blk * (500000 // period + 2))[:500000]
        for level in (1, 3, 6):
            comp = _enc(data, level=level)
            assert zlib.decompress(comp, -15) == data, (period, level)
    # (one position per slot: a copy is found if the slot has not been taken again since -- 2 000 positions back, yes)
    noise = rnd.randbytes(2000)
    once, many = len(_enc(noise, level=1)), len(_enc(noise * 100, level=1))  # 200 000 bytes: the copies lie on both sides of 64 KiB marks
    assert many < once + 99 * 150, (once, many)


def test_block_kinds_and_levels(alice):
    def first_btype(comp):
        return (comp[0] >> 1) & 3

    assert first_btype(_enc(alice, level=0)) == 0  # stored
    assert first_btype(_enc(alice, level=1)) == 1  # one fixed-Huffman block (BASELINE.json configs[3])
    assert first_btype(_enc(alice, level=6)) == 2  # dynamic
    assert first_btype(_enc(alice, level=6, strategy=4)) == 1  # Z_FIXED
    assert first_btype(_enc(os.urandom(50000), level=6)) == 0  # incompressible: stored is cheapest
    assert first_btype(_enc(b"ab", level=6)) == 1  # tiny: the fixed code beats a dynamic header
    l1, l3, l6 = len(_enc(alice, level=1)), len(_enc(alice, level=3)), len(_enc(alice, level=6))
    assert l6 < l3 < l1  # dynamic codes, then lazy choice
    # levels share match finders in three groups: greedy (2-3), lazy (4-5), lazy with two positions per hash slot (6-9)
    assert _enc(alice, level=2) == _enc(alice, level=3) and _enc(alice, level=4) == _enc(alice, level=5)
    assert _enc(alice, level=6) == _enc(alice, level=9) and len(_enc(alice, level=6)) < 0.96 * len(_enc(alice, level=4))
    # HuffmanOnly: no matches at all; Rle: only distance 1
    assert len(_enc(b"abc" * 5000, level=6, strategy=2)) > 3000 > len(_enc(b"abc" * 5000, level=6)) > 0
    assert len(_enc(b"\0" * 100000, level=6, strategy=3)) < 600


def test_stated_ratio_bounds(alice):
    from bench_support import synth

    for data in (alice, synth.payloads(4).tobytes()):
        z1 = len(zlib.compress(data, 1)) - 6
        assert len(_enc(data, level=6)) <= 1.15 * z1  # the default level
        assert len(_enc(data, level=1)) <= 1.5 * z1   # level 1: fixed Huffman by specification
    assert len(_enc(alice, level=6)) <= len(zlib.compress(alice, 1)) - 6  # text: lazy + dynamic beats zlib level 1


@pytest.mark.parametrize("level", [1, 6])
def test_status_contract_with_small_outputs(alice, level):
    """Process takes input (Continue), Flush makes everything so far decodable, Finish with a small output says NeedOutput
    until the stream has been handed out (src/encoder/mod.rs:357-367; tests/encoder.rs:115-173 for the empty Finish)."""
    rnd = random.Random(level)
    for mode in WBITS:
        e = O.DeflateEncoder(mode, level, 0)
        data = (alice * 2)[: rnd.randrange(100000, 300000)]
        out, pos = b"", 0
        d = zlib.decompressobj(WBITS[mode])
        seen = b""
        while pos < len(data):
            piece = data[pos : pos + rnd.randrange(1, 50000)]
            pos += len(piece)
            op = rnd.choice([O.OP_PROCESS, O.OP_FLUSH])
            o, ir, orr, st = e.encode(piece, 1 << 20, op)
            assert ir == 0 and st == O.ENC_CONTINUE
            out += o
            if op == O.OP_FLUSH:
                seen += d.decompress(o)
                assert seen == data[:pos]  # a sync flush point: all input so far comes out
            else:
                assert o == b""  # nothing is produced before a flush in this encoder
        steps = 0
        while True:
            o, ir, orr, st = e.encode(b"", 777, O.OP_FINISH)
            out += o
            steps += 1
            if st == O.ENC_FINISHED:
                break
            assert st == O.ENC_NEED_OUTPUT and len(o) == 777
        assert steps > 1 and zlib.decompress(out, WBITS[mode]) == data
        o, ir, orr, st = e.encode(b"", 100, O.OP_FINISH)  # finishing again: nothing more, still Finished
        assert (o, st) == (b"", O.ENC_FINISHED)
        e.reset()
        o, ir, orr, st = e.encode(b"", 100, O.OP_FINISH)  # empty stream after reset
        assert st == O.ENC_FINISHED and zlib.decompress(o, WBITS[mode]) == b""
