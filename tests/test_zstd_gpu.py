"""zstd frame decode on the GPU: the reference's decoder test (tests/decoder.rs:119-128,
should_decode_zstd) against the hip variant, and batch parity with the oracle."""
import random

import numpy as np
import pytest

import zstd_ref
from conftest import golden
from test_decoder_gpu import DATA, _test_case
from test_inflate_gpu import _mk, run_batch

pytestmark = pytest.mark.gpu
FMT_ZSTD = 100


def oracle_zstd_batch(parts, caps):
    from oracle import oracle as O

    res = []
    for p, c in zip(parts, caps):
        got, ir, orr, st, err = O.ZstdDecoder().decode(p, int(c))
        res.append((got, len(p) - ir, err if err else st))
    return res


def test_should_decode_zstd_hip(gpu):
    """tests/decoder.rs:119-128 with Interface::zstd replaced by the hip variant"""
    import compu_amd

    decoder = compu_amd.decoder_interface.zstd_hip(compu_amd.ZstdOptions())
    assert decoder is not None, "create zstd-hip decoder"
    for name in DATA:
        _test_case(compu_amd, decoder, golden(name), golden(name + ".compressed.zstd"))
    assert decoder.describe_error(compu_amd.DecodeError.no_error()) == "No error detected"


def test_zstd_fixture_error_classes(gpu, alice):
    import compu_amd

    comp = golden("alice29.txt.compressed.zstd")
    dec = compu_amd.decoder_interface.zstd_hip()
    out = bytearray(len(alice) + 16)
    bad = bytearray(comp)
    bad[-1] ^= 1  # checksum mismatch -> Err(-22)
    r = dec.decode(bytes(bad), out)
    assert not r.is_ok() and r.status.as_raw() == -22
    assert dec.describe_error(r.status) == "Restored data doesn't match checksum"
    dec.reset()
    r = dec.decode(b"\x00\x01\x02\x03\x04\x05", out)  # not a zstd frame -> Err(-10)
    assert not r.is_ok() and r.status.as_raw() == -10
    dec.reset()
    r = dec.decode(comp[: len(comp) // 2], out)  # first half: the complete blocks come out, then NeedInput
    assert r.status == compu_amd.DecodeStatus.NeedInput and r.input_remain == 0
    n1 = len(out) - r.output_remain
    assert bytes(out[:n1]) == alice[:n1] and n1 > 0
    r = dec.decode(comp[len(comp) // 2 :] + b"xy", out, n1, len(out) - n1)
    assert r.status == compu_amd.DecodeStatus.Finished and r.input_remain == 2
    assert bytes(out[: len(alice)]) == alice
    # window_log cap (ZstdOptions::window_log, src/decoder/zstd.rs:36-47): the fixture needs 152089 > 2^17
    small = compu_amd.decoder_interface.zstd_hip(compu_amd.ZstdOptions().window_log(17))
    r = small.decode(comp, out)
    assert not r.is_ok() and r.status.as_raw() == -16


def test_generated_frames_batch_match_oracle(gpu, alice):
    z = zstd_ref.load()
    assert z is not None
    rnd = random.Random(8)
    datas, parts = [], []
    for it in range(250):
        n = rnd.choice([0, 1, 2, 10, 100, 1000, 5000, 65536, 70000, 140000])
        data = _mk(rnd.randrange(5), n, rnd, alice)
        comp = zstd_ref.compress(z, data, rnd.choice([1, 3, 5, 9, 15, 19, -1]), rnd.random() < 0.7, rnd.random() < 0.8)
        if it % 10 == 0:
            comp = b"\x50\x2a\x4d\x18\x03\x00\x00\x00abc"  # a skippable frame: finished with no output
            data = b""
        datas.append(data)
        parts.append(comp + (b"tail" if it % 3 == 0 else b""))
    caps = [len(d) + rnd.choice([0, 0, 9, 200]) for d in datas]
    outs, ol, iu, st = run_batch(gpu, FMT_ZSTD, parts, [max(c, 1) for c in caps], check_tail=False)
    ref = oracle_zstd_batch(parts, [max(c, 1) for c in caps])
    for i in range(len(parts)):
        assert st[i] == 2 == ref[i][2], (i, st[i], ref[i][2])
        assert outs[i] == datas[i] == ref[i][0], i
        assert iu[i] == ref[i][1], (i, iu[i], ref[i][1])


def test_truncated_and_corrupt_frames_match_oracle(gpu, alice):
    z = zstd_ref.load()
    rnd = random.Random(12)
    parts, caps = [], []
    for it in range(600):
        n = rnd.choice([50, 500, 5000, 70000])
        data = _mk(rnd.choice([1, 2, 4, 0]), n, rnd, alice)
        comp = bytearray(zstd_ref.compress(z, data, rnd.choice([1, 3, 9]), rnd.random() < 0.7, rnd.random() < 0.8))
        mode = rnd.randrange(3)
        if mode == 0:
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            comp = comp[: rnd.randrange(len(comp))]
        parts.append(bytes(comp))
        caps.append(n + 300)
    outs, ol, iu, st = run_batch(gpu, FMT_ZSTD, parts, caps, check_tail=False)
    ref = oracle_zstd_batch(parts, caps)
    for i in range(len(parts)):
        r_out, r_used, r_st = ref[i]
        if r_st == 1:
            # The (corrupted) frame regenerates more than the capacity.  The streaming oracle stops the moment the output is
            # full; the batch kernel works on whole blocks (include/compu_hip.h, chip_decode_batch): it either reports the
            # same NeedOutput, with the whole blocks that fit, or -- when the block that does not fit is itself broken --
            # the verdict the oracle reaches on that frame once capacity is no object.  Nothing else.
            big = oracle_zstd_batch([parts[i]], [caps[i] + (1 << 20)])[0]
            assert st[i] == 1 or (big[2] < 0 and st[i] == big[2]), (i, st[i], big[2])
            assert len(outs[i]) <= caps[i] and outs[i] == r_out[: len(outs[i])]
            continue
        assert st[i] == r_st, (i, st[i], r_st)
        if r_st in (0, 2):
            assert outs[i] == r_out, (i, len(outs[i]), len(r_out))
        if r_st == 2:
            assert iu[i] == r_used


def test_compu_status_flag_puts_a_full_output_first(gpu, alice):
    """CHIP_F_COMPU_STATUS for zstd (src/decoder/zstd.rs:121-133): compu compares output.pos with output.size first.  A truncated frame
    whose blocks so far fill the range is NeedOutput, and so is anything handed an empty range; an ERROR stays the error -- also behind
    an exactly sized buffer, because ZSTD_decompressStream returns it before it writes output.pos (compu sees 0).  The expectations are
    the system libzstd's own answers (zstd_ref.stream_decode_once), the oracle's, and the kernel's."""
    import compu_amd

    z = zstd_ref.load()
    data = alice[:100000]
    good = zstd_ref.compress(z, data, 3, True, True)
    bad_ck = bytearray(good)
    bad_ck[-1] ^= 1
    two = zstd_ref.compress(z, (alice * 3)[:400000], 3, True, False)  # four blocks
    parts = [bytes(bad_ck), bytes(bad_ck), good, good[:-2], b"\x00\x00\x00\x00", good[:40], bytes(bad_ck)]
    caps = [100000, 100001, 100000, 100000, 0, 0, 0]
    outs, ol, iu, st = run_batch(gpu, FMT_ZSTD, parts, caps, check_tail=False, flags=compu_amd.F_COMPU_STATUS)
    ref = oracle_zstd_batch(parts, caps)
    real = []
    for p_, c_ in zip(parts, caps):
        _o, _i, _r, s_, e_ = zstd_ref.stream_decode_once(z, p_, c_)
        real.append(e_ if s_ is None else s_)
    assert [int(x) for x in st] == [r[2] for r in ref] == real == [-22, -22, 2, 1, 1, 1, 1]
    outs0, _, _, st0 = run_batch(gpu, FMT_ZSTD, parts, caps, check_tail=False)
    assert [int(x) for x in st0] == [-22, -22, 2, 0, -10, 0, 1]  # the default names the cause (no room at all for the last one)
    # a truncated multi-block frame: the whole blocks in front of the cut fill the buffer exactly
    full, _, _, s_full = run_batch(gpu, FMT_ZSTD, [two], [400000], check_tail=False)
    assert s_full[0] == 2
    cut = two[: len(two) * 9 // 10]
    o, l, _, s = run_batch(gpu, FMT_ZSTD, [cut], [400000], check_tail=False)
    assert s[0] == 0 and 0 < l[0] < 400000
    o2, l2, _, s2 = run_batch(gpu, FMT_ZSTD, [cut, cut], [int(l[0]), int(l[0]) + 1], check_tail=False, flags=compu_amd.F_COMPU_STATUS)
    r2 = oracle_zstd_batch([cut, cut], [int(l[0]), int(l[0]) + 1])
    assert [int(x) for x in s2] == [r[2] for r in r2] == [1, 0] and o2[0] == o2[1] == o[0]


def test_mixed_gzip_zstd_batch_routed_by_detection(gpu, alice):
    """BASELINE.json configs[4] in small: a batch of gzip, zlib and zstd units routed per unit by
    Detection::detect (src/decoder/mod.rs:28-114); unknown units are reported, not decoded."""
    import zlib

    import compu_amd

    z = zstd_ref.load()
    rnd = random.Random(77)
    datas, parts, kinds = [], [], []
    for it in range(300):
        n = rnd.choice([0, 10, 1000, 65536])
        data = _mk(rnd.randrange(5), n, rnd, alice)
        k = rnd.randrange(5)
        if k == 0:
            comp = zstd_ref.compress(z, data, 3, True, True)
        elif k == 1:
            co = zlib.compressobj(6, zlib.DEFLATED, 31)
            comp = co.compress(data) + co.flush()
        elif k == 2:
            comp = zlib.compress(data, 6)  # 78 9c
        elif k == 3:
            co = zlib.compressobj(6, zlib.DEFLATED, 13)  # CINFO 5 -> 58 85: a zlib header the reference's table knows
            comp = co.compress(data) + co.flush()
        else:
            comp = b"PK\x03\x04 not a known stream"
        datas.append(data)
        parts.append(comp)
        kinds.append(k)
    outs, ol, iu, st = run_batch(gpu, 0, parts, [max(len(d), 1) for d in datas], check_tail=False)
    for i, k in enumerate(kinds):
        assert compu_amd.Detection.detect(parts[i]) == [compu_amd.Detection.Zstd, compu_amd.Detection.Gzip, compu_amd.Detection.Zlib,
                                                        compu_amd.Detection.Zlib, compu_amd.Detection.Unknown][k]
        if k == 4:
            assert st[i] == 4 and ol[i] == 0  # CHIP_UNKNOWN_FORMAT
        else:
            assert st[i] == 2 and outs[i] == datas[i] and iu[i] == len(parts[i]), (i, k, st[i])
    # the 0x68 quirk of the reference table (mod.rs:80-82): a valid `68 81` zlib stream is Unknown to the router
    co = zlib.compressobj(6, zlib.DEFLATED, 14)
    q = co.compress(b"quirk") + co.flush()
    assert q[:2] == b"\x68\x81" and compu_amd.Detection.detect(q) == compu_amd.Detection.Unknown
    outs, ol, iu, st = run_batch(gpu, 0, [q], [16], check_tail=False)
    assert st[0] == 4


def test_large_multiblock_frames(gpu, alice):
    """Frames of several MB: many 128 KiB blocks, repeat-mode tables, treeless literals, matches far beyond
    one block (window up to 8 MiB at level 19), raw and RLE blocks."""
    z = zstd_ref.load()
    rnd = random.Random(5)
    big = bytearray()
    while len(big) < 5_000_000:
        k = rnd.randrange(5)
        if k == 0:
            big += alice[rnd.randrange(len(alice) // 2) :][: rnd.randrange(1, 200000)]
        elif k == 1:
            big += rnd.randbytes(rnd.randrange(1, 200000))
        elif k == 2:
            big += b"\0" * rnd.randrange(1, 400000)
        elif k == 3:
            big += bytes(rnd.choice(b"abc") for _ in range(rnd.randrange(1, 20000)))
        else:
            big += big[-rnd.randrange(1, len(big) + 1) :][: rnd.randrange(1, 50000)] if big else b"x"
    big = bytes(big)
    for level in (1, 3, 19):
        comp = zstd_ref.compress(z, big, level, True, True)
        outs, ol, iu, st = run_batch(gpu, FMT_ZSTD, [comp], [len(big)], check_tail=False)
        assert st[0] == 2 and iu[0] == len(comp) and outs[0] == big, (level, st[0], ol[0])


def test_streaming_a_long_zstd_frame_in_small_pieces_is_linear(gpu, alice):
    """A multi-block frame fed 16 KiB at a time: every call continues from the last completed block (the kernel's
    checkpoint: cursor, repeat offsets, decode tables) instead of decoding the frame again from its header."""
    import time

    import compu_amd as compu

    z = zstd_ref.load()
    rnd = random.Random(31)
    big = bytearray()
    while len(big) < 6_000_000:
        k = rnd.randrange(4)
        if k == 0:
            big += alice[rnd.randrange(len(alice) // 2) :][: rnd.randrange(1, 100000)]
        elif k == 1:
            big += rnd.randbytes(rnd.randrange(1, 30000))
        elif k == 2:
            big += bytes([rnd.randrange(256)]) * rnd.randrange(1, 50000)
        else:
            big += big[-rnd.randrange(1, min(len(big), 100000) + 1) :][: rnd.randrange(1, 3000)] if big else b"x"
    big = bytes(big)
    comp = zstd_ref.compress(z, big, 3, True, True)
    dec = compu.decoder_interface.zstd_hip()
    out = bytearray()
    buf = bytearray(1 << 20)
    t0 = time.perf_counter()
    pos = 0
    calls = 0
    while True:
        chunk = comp[pos : pos + 16384]
        r = dec.decode(chunk, buf)
        calls += 1
        assert r.is_ok(), r.status
        out += buf[: len(buf) - r.output_remain]
        pos += len(chunk) - r.input_remain
        if r.status == compu.DecodeStatus.Finished:
            break
        if r.status == compu.DecodeStatus.NeedInput:
            assert pos < len(comp)
    dt = time.perf_counter() - t0
    assert bytes(out) == big and pos == len(comp)
    assert dt < 60, f"{calls} calls took {dt:.1f} s: the frame is being decoded from its start again"
    # the same decoder, reset, decodes another frame from its header
    dec.reset()
    small = zstd_ref.compress(z, alice[:5000], 3, True, True)
    o2 = bytearray(6000)
    r = dec.decode(small, o2)
    assert r.status == compu.DecodeStatus.Finished and bytes(o2[: len(o2) - r.output_remain]) == alice[:5000]


def _compress_windowed(z, data, level, window_log, ldm=False, content_size=True):
    """One frame with an explicit window (ZSTD_c_windowLog): larger inputs get a windowed, not a single-segment, frame."""
    import ctypes as C

    cctx = z.ZSTD_createCCtx()
    z.ZSTD_CCtx_setParameter(cctx, 100, level)
    z.ZSTD_CCtx_setParameter(cctx, 101, window_log)  # ZSTD_c_windowLog
    z.ZSTD_CCtx_setParameter(cctx, 201, 1)  # ZSTD_c_checksumFlag
    z.ZSTD_CCtx_setParameter(cctx, 200, 1 if content_size else 0)
    if ldm:
        z.ZSTD_CCtx_setParameter(cctx, 160, 1)  # ZSTD_c_enableLongDistanceMatching
    cap = z.ZSTD_compressBound(len(data))
    dst = (C.c_char * cap)()
    src = (C.c_char * len(data)).from_buffer(data) if isinstance(data, bytearray) else C.create_string_buffer(bytes(data), len(data))
    n = z.ZSTD_compress2(cctx, dst, cap, src, len(data))
    z.ZSTD_freeCCtx(cctx)
    assert not z.ZSTD_isError(n)
    return bytes(memoryview(dst)[:n])


def _stream_through(compu, dec, comp, piece, out_cap, gpu=None, every=64):
    """compu's decode loop (src/decoder/mod.rs:323-335) over `comp`; -> (crc32 of the output, its length, peaks)"""
    import zlib

    out = bytearray(out_cap)
    crc, got, pos, calls = 0, 0, 0, 0
    peak_pin = peak_dev = 0
    min_free = gpu.cuda.mem_get_info()[0] if gpu else 0
    status = None
    while status != compu.DecodeStatus.Finished:
        chunk = comp[pos : pos + piece]
        r = dec.decode(chunk, out)
        assert r.is_ok(), (pos, r.status)
        n = len(out) - r.output_remain
        crc = zlib.crc32(memoryview(out)[:n], crc)
        got += n
        pos += len(chunk) - r.input_remain
        status = r.status
        calls += 1
        if gpu and calls % every == 0:
            pin, devb = dec.footprint()
            peak_pin, peak_dev = max(peak_pin, pin), max(peak_dev, devb)
            min_free = min(min_free, gpu.cuda.mem_get_info()[0])
        assert calls < 400000
    assert pos == len(comp)
    return crc, got, peak_pin, peak_dev, min_free


def test_streaming_memory_stays_bounded_over_a_long_zstd_frame(gpu, alice):
    """A 192 MiB windowed frame (1 MiB window, content size and checksum in the frame) fed in 64 KiB pieces: the decoder drops
    input in front of the last block and output behind the window that has been handed on, and carries the XXH64 state from
    block to block -- O(window + piece), as ZSTD_decompressStream holds it (src/decoder/zstd.rs:98-148)."""
    import zlib

    import compu_amd as compu

    z = zstd_ref.load()
    rnd = random.Random(78)
    total = 192 << 20
    data = bytearray()
    while len(data) < total:
        kind = rnd.randrange(3)
        if kind == 0:
            s0 = rnd.randrange(0, len(alice) - 65536)
            data += (alice[s0 : s0 + 65536]) * 16
        elif kind == 1:
            data += rnd.randbytes(1 << 20)
        else:
            data += bytes([rnd.randrange(256)]) * (1 << 20)
    want_crc = zlib.crc32(data)
    comp = _compress_windowed(z, data, 1, 20)
    made = len(data)
    del data
    dec = compu.decoder_interface.zstd_hip()
    free0 = gpu.cuda.mem_get_info()[0]
    crc, got, peak_pin, peak_dev, min_free = _stream_through(compu, dec, comp, 64 << 10, 256 << 10, gpu)
    assert got == made and crc == want_crc
    assert peak_pin <= 4 << 20 and peak_dev <= 24 << 20, (peak_pin, peak_dev)
    assert free0 - min_free <= 64 << 20, (free0, min_free)  # hipMemGetInfo: no growth with the frame's length
    # a corrupted checksum at the very end is still found (the running state, not a re-read of the output, decides)
    dec.reset()
    bad = bytearray(comp)
    bad[-1] ^= 0x55
    out = bytearray(1 << 20)
    pos, status = 0, None
    while True:
        chunk = bytes(bad[pos : pos + (1 << 20)])
        r = dec.decode(chunk, out)
        if not r.is_ok():
            status = r.status
            break
        pos += len(chunk) - r.input_remain
        assert r.status != compu.DecodeStatus.Finished
    assert status == compu.DecodeError(-22)  # ZSTD_error_checksum_wrong


def test_streaming_a_zstd_window_larger_than_the_device_buffer(gpu, alice):
    """A 16 MiB window with matches that reach 12 MiB back (long-distance matching): the device output grows to hold the window
    (the 8 MiB soft limit gives way when a run makes no progress), the bytes are right, and a frame without a content size works
    the same."""
    import zlib

    import compu_amd as compu

    z = zstd_ref.load()
    rnd = random.Random(79)
    a = bytearray()
    while len(a) < (12 << 20):
        s0 = rnd.randrange(0, len(alice) - 4096)
        a += alice[s0 : s0 + rnd.randrange(64, 4096)] + rnd.randbytes(rnd.randrange(16, 512))
    data = bytearray(a) + a[: 6 << 20] + bytearray(rnd.randbytes(1 << 20)) + a[3 << 20 : 9 << 20]
    want_crc = zlib.crc32(data)
    for content_size in (True, False):
        comp = _compress_windowed(z, data, 3, 24, ldm=True, content_size=content_size)
        assert len(comp) < len(data) * 0.7  # the far copies were found: offsets beyond 8 MiB are in the frame
        dec = compu.decoder_interface.zstd_hip()
        crc, got, _, peak_dev, _ = _stream_through(compu, dec, comp, 256 << 10, 1 << 20, gpu, every=8)
        assert got == len(data) and crc == want_crc
        assert (16 << 20) <= peak_dev <= (72 << 20), peak_dev


def test_an_offset_beyond_the_window_is_corruption_batch_and_stream(gpu):
    """an offset that reaches behind the frame's window -- but not behind its start -- is corruption in the batch
    path and in the streaming path alike, before and after the streaming decoder has let history go (it keeps the window; it used to
    accept such an offset while the bytes happened to be there and reject it after 1 MiB had been dropped).  Frames written by hand
    (tests/zstd_ref.py::craft_offset_frame; window 1 KiB), verdicts and bytes against the oracle."""
    import zlib

    import compu_amd as compu

    import torch

    cases = [(2, 1000), (2, 1024), (2, 1025), (2, 1500), (2, 2000), (2, 2001), (1, 900), (1, 1001), (40, 1024), (40, 30000)]
    frames = [zstd_ref.craft_offset_frame(b, 1000, off)[0] for b, off in cases]
    caps = [b * 1000 + 64 for b, _ in cases]
    outs, ol, iu, st = run_batch(torch, FMT_ZSTD, frames, caps, check_tail=False)
    ref = oracle_zstd_batch(frames, caps)
    for j, (b, off) in enumerate(cases):
        r_out, r_used, r_st = ref[j]
        assert r_st == (2 if off <= min(1024, b * 1000) else -20)
        assert int(st[j]) == r_st and (r_st != 2 or (outs[j] == r_out and int(iu[j]) == r_used)), (b, off, int(st[j]), r_st)
    # streaming: 1.6 MB of raw blocks in front of the sequence, so that the decoder has dropped more than 1 MiB by then
    for off, ok in ((1024, True), (1025, False), (1500, False), (700000, False)):
        frame, data = zstd_ref.craft_offset_frame(1600, 1000, off)
        for piece, room in ((64 << 10, 64 << 10), (len(frame), 2 << 20), (5000, 300000)):
            dec = compu.decoder_interface.zstd_hip()
            out = bytearray(room)
            pos, crc, got, err = 0, 0, 0, None
            for _ in range(100000):
                chunk = frame[pos : pos + piece]
                r = dec.decode(chunk, out)
                if not r.is_ok():
                    err = r.status.as_raw()
                    break
                n = len(out) - r.output_remain
                crc = zlib.crc32(memoryview(out)[:n], crc)
                got += n
                pos += len(chunk) - r.input_remain
                if r.status == compu.DecodeStatus.Finished:
                    break
            if ok:
                want = data + data[len(data) - off : len(data) - off + 3]
                assert err is None and got == len(want) and crc == zlib.crc32(want), (off, piece, err, got)
            else:
                assert err == -20, (off, piece, err, got)


def test_literal_paths_rows_and_second_decode(gpu):
    """The literal decoder keeps the walk's symbols in scratch rows inside the frame's own output range and copies the owned pieces
    out of them; it decodes a second time instead when the range has no room for the rows (small frames, tight capacities) or when a
    round has lanes that walk more than 32 trips (codes of one or two bits: up to 256 symbols in a 256-bit segment).  Frames that take
    each way -- skewed alphabets with a one-bit code, flat ones, sizes around the rows' 8.7 KB, capacity exactly the content size --
    against the oracle."""
    z = zstd_ref.load()
    assert z is not None
    rnd = random.Random(21)
    datas, parts = [], []
    for it in range(60):
        n = rnd.choice([3000, 9000, 12000, 40000, 65536, 131072, 200000])
        kind = it % 4
        if kind == 0:    # one dominant byte: a one-bit code, more than 128 symbols per lane and round
            data = bytes(97 if rnd.random() < 0.88 else rnd.randrange(256) for _ in range(n))
        elif kind == 1:  # two dominant bytes: two-bit codes
            data = bytes(rnd.choice(b"ab") if rnd.random() < 0.9 else rnd.randrange(256) for _ in range(n))
        elif kind == 2:  # flat over 64 symbols: six-bit codes, the rows' usual case
            data = bytes(32 + rnd.randrange(64) for _ in range(n))
        else:            # a mix of both in one frame (several blocks when n > 128 KiB)
            half = n // 2
            data = bytes(97 if rnd.random() < 0.9 else rnd.randrange(256) for _ in range(half)) + bytes(32 + rnd.randrange(64) for _ in range(n - half))
        comp = zstd_ref.compress(z, data, rnd.choice([1, 3]), True, rnd.random() < 0.7)
        datas.append(data)
        parts.append(comp)
    caps = [len(d) + (0 if i % 2 else 4096) for i, d in enumerate(datas)]
    outs, ol, iu, st = run_batch(gpu, FMT_ZSTD, parts, caps, check_tail=False)
    ref = oracle_zstd_batch(parts, caps)
    for i in range(len(parts)):
        assert st[i] == 2 == ref[i][2], (i, st[i], ref[i][2])
        assert outs[i] == datas[i] == ref[i][0], i
        assert iu[i] == ref[i][1], (i, iu[i], ref[i][1])
