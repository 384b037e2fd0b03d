"""The reference's own decoder test (tests/decoder.rs:21-77, should_decode_zlib_ng_gzip :141-150)
replayed phase by phase against the `hip` Interface variant, plus wrapper/trailer parity with the oracle."""
import random
import zlib

import numpy as np
import pytest

from conftest import golden
from test_inflate_gpu import _mk, oracle_batch, run_batch

pytestmark = pytest.mark.gpu

DATA = ["10x10y", "alice29.txt"]


def _test_case(compu, decoder, data, compressed):
    """tests/decoder.rs:21-77"""
    DecodeStatus, DecodeError = compu.DecodeStatus, compu.DecodeError
    # Full (:25-31)
    output = bytearray(len(data))
    result = decoder.decode(compressed, output)
    assert result.status == DecodeStatus.Finished
    assert result.input_remain == 0
    assert result.output_remain == 0
    assert bytes(output) == data
    decoder.reset()

    # Partial buffer (:34-44); DATA.len() / 2 == 1 in the reference (DATA is the 2-element array)
    half = len(DATA) // 2
    result = decoder.decode(compressed, output, 0, half)
    assert result.status == DecodeStatus.NeedOutput
    assert result.output_remain == 0
    remaining = compressed[len(compressed) - result.input_remain :]
    result = decoder.decode(remaining, output, half, len(output) - half)
    assert result.status == DecodeStatus.Finished
    assert bytes(output) == data
    decoder.reset()

    # Buffered decoder (:47-63)
    buffer = compu.Buffer(4096)
    buffer_input = compressed
    out = bytearray()
    while True:
        consumed, status = buffer.decode(decoder, buffer_input)
        buffer_input = buffer_input[consumed:]
        out += buffer.data()
        buffer.consume()
        if status == DecodeStatus.Finished:
            break
    assert bytes(out) == data
    decoder.reset()

    # Full vec (:66-72)
    vec = compu.Vec()
    result = decoder.decode_vec_full(compressed, vec)
    assert result.status == DecodeStatus.Finished
    assert result.input_remain == 0
    assert bytes(vec) == data
    decoder.reset()

    # :74-76
    assert decoder.describe_error(DecodeError.no_error()) is not None


def test_should_decode_zlib_hip_gzip(gpu):
    """tests/decoder.rs:141-150 with Interface::zlib_ng replaced by the hip variant"""
    import compu_amd

    decoder = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
    assert decoder is not None, "create zlib-hip decoder"
    for name in DATA:
        _test_case(compu_amd, decoder, golden(name), golden(name + ".compressed.gz"))
    # the default mode (Auto, zlib_common.rs:24-29) takes the same fixtures
    decoder = compu_amd.decoder_interface.zlib_hip()
    for name in DATA:
        _test_case(compu_amd, decoder, golden(name), golden(name + ".compressed.gz"))


def test_stream_triples_match_oracle(gpu, alice):
    """status / output_remain classes of SURVEY.md sec. 8b on chunked input, small outputs, truncation, corruption"""
    import compu_amd
    from oracle import oracle as O

    comp = golden("alice29.txt.compressed.gz")
    dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
    # first quarter of the input, big output -> NeedInput, all input consumed
    out = bytearray(len(alice))
    r = dec.decode(comp[: len(comp) // 4], out)
    o = O.InflateDecoder(O.MODE_GZIP)
    ref, ir, orr, st, err = o.decode(comp[: len(comp) // 4], len(alice))
    assert r.status == compu_amd.DecodeStatus.NeedInput and r.input_remain == 0 == ir
    assert r.output_remain == orr and bytes(out[: len(alice) - r.output_remain]) == ref
    # the rest in 4 KiB chunks (the shape of tests/encoder.rs:152-171)
    pos = len(comp) // 4
    got = bytearray(ref)
    while True:
        chunk = comp[pos : pos + 4096]
        buf = bytearray(len(alice))
        r = dec.decode(chunk, buf)
        got += buf[: len(buf) - r.output_remain]
        pos += len(chunk) - r.input_remain
        assert r.is_ok()
        if r.status == compu_amd.DecodeStatus.Finished:
            break
        assert r.status == compu_amd.DecodeStatus.NeedInput and r.input_remain == 0
    assert bytes(got) == alice and pos == len(comp)
    dec.reset()
    # truncated by one byte: everything is decoded, the trailer is incomplete -> NeedInput
    r = dec.decode(comp[:-1], out)
    assert r.status == compu_amd.DecodeStatus.NeedInput and bytes(out) == alice and r.output_remain == 0
    dec.reset()
    # corrupted payload -> Err(-3) "data error"
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x10
    r = dec.decode(bytes(bad), out)  # decodes to one byte more than the original: room first
    ref, ir, orr, st, err = O.InflateDecoder(O.MODE_GZIP).decode(bytes(bad), len(alice))
    assert r.status == compu_amd.DecodeStatus.NeedOutput == st and r.output_remain == 0 == orr and bytes(out) == ref
    dec.reset()
    big = bytearray(2 * len(alice))
    r = dec.decode(bytes(bad), big)
    ref, ir, orr, st, err = O.InflateDecoder(O.MODE_GZIP).decode(bytes(bad), len(big))
    assert not r.is_ok() and r.status.as_raw() == err == -3
    assert r.output_remain == orr and bytes(big[: len(big) - orr]) == ref
    assert dec.describe_error(r.status) == "data error"
    dec.reset()
    # trailing bytes after the stream stay in input_remain
    r = dec.decode(comp + b"tail!", out)
    assert r.status == compu_amd.DecodeStatus.Finished and r.input_remain == 5 and bytes(out) == alice
    # a finished decoder stays finished until reset (README.md:43-44)
    r = dec.decode(b"xyz", out)
    assert r.status == compu_amd.DecodeStatus.Finished and r.input_remain == 3


def test_wrapped_units_batch_match_oracle(gpu, alice):
    """zlib / gzip / auto wrappers incl. FEXTRA/FNAME/FCOMMENT/FHCRC headers, bad trailers, FDICT"""
    import gzip
    import io
    import struct

    rnd = random.Random(21)

    def gz_with_fields(data, flags):
        raw = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = raw.compress(data) + raw.flush()
        hdr = bytearray(b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\x02\x03")
        if flags & 4:
            extra = rnd.randbytes(rnd.randrange(0, 300))
            hdr += struct.pack("<H", len(extra)) + extra
        if flags & 8:
            hdr += bytes(rnd.randrange(1, 256) for _ in range(rnd.randrange(0, 200))) + b"\0"
        if flags & 16:
            hdr += bytes(rnd.randrange(1, 256) for _ in range(rnd.randrange(0, 200))) + b"\0"
        if flags & 2:
            hdr += struct.pack("<H", zlib.crc32(bytes(hdr)) & 0xFFFF)
        return bytes(hdr) + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)

    for fmt in (15, 31, 47):
        datas, parts = [], []
        for it in range(150):
            n = rnd.choice([0, 1, 20, 300, 5000, 65536])
            data = _mk(rnd.choice([0, 1, 2, 3, 4]), n, rnd, alice)
            kind = rnd.randrange(6)
            wb = fmt if fmt != 47 else rnd.choice([15, 31])
            if wb == 31 and kind == 0:
                comp = gz_with_fields(data, rnd.randrange(32) & 0x1E)
            else:
                co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, wb)
                comp = co.compress(data) + co.flush()
            comp = bytearray(comp)
            if kind == 1:  # break the trailer
                comp[-rnd.randrange(1, 5 if wb == 15 else 9)] ^= 0x01
            elif kind == 2:  # truncate somewhere (header, body or trailer)
                comp = comp[: rnd.randrange(len(comp))]
            elif kind == 3:  # damage a header byte
                comp[rnd.randrange(min(len(comp), 4))] ^= 1 << rnd.randrange(8)
            elif kind == 4:
                comp += b"trailing"
            datas.append(data)
            parts.append(bytes(comp))
        caps = [max(len(d), 1) for d in datas]
        outs, ol, iu, st = run_batch(gpu, fmt, parts, caps)
        ref = oracle_batch(fmt, parts, caps)
        for i in range(len(parts)):
            r_out, r_used, r_st = ref[i]
            if len(parts[i]) == 0 and st[i] == 0 and r_st == 1:
                continue
            if st[i] == 1 and r_st == 0 and len(r_out) == caps[i]:
                continue
            assert st[i] == r_st, (fmt, i, st[i], r_st)
            assert outs[i] == r_out, (fmt, i)
            if r_st == 2:
                assert iu[i] == r_used, (fmt, i)
    # FDICT zlib header -> Z_NEED_DICT (2) once the dictionary id is there
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY, b"dictionary")
    comp = co.compress(b"dictionary words") + co.flush()
    outs, ol, iu, st = run_batch(gpu, 15, [comp, comp[:4]], [100, 100])
    ref = oracle_batch(15, [comp, comp[:4]], [100, 100])
    # oracle_batch folds err=2 (Z_NEED_DICT) into its status slot; the batch API calls it CHIP_NEED_DICT = 3
    assert st[0] == 3 and ref[0][2] == 2 and ref[0][0] == b"" and st[1] == 0 == ref[1][2]
    import compu_amd

    dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Zlib)
    r = dec.decode(comp, bytearray(100))
    assert not r.is_ok() and r.status.as_raw() == 2 and dec.describe_error(r.status) == "need dictionary"


def test_streaming_a_long_gzip_in_small_pieces_is_linear(gpu, alice):
    """A long stream fed 8 KiB at a time: every call continues from the last block boundary reached (the kernel's
    resume state) instead of decoding the stream again from its start -- a thousand calls stay cheap -- and the
    result is the same as in one piece, including the gzip trailer check at the end."""
    import random
    import time
    import zlib

    import compu_amd as compu

    rnd = random.Random(8)
    big = bytearray()
    while len(big) < 8_000_000:
        k = rnd.randrange(4)
        if k == 0:
            big += alice[rnd.randrange(len(alice) // 2) :][: rnd.randrange(1, 100000)]
        elif k == 1:
            big += rnd.randbytes(rnd.randrange(1, 30000))
        elif k == 2:
            big += bytes([rnd.randrange(256)]) * rnd.randrange(1, 50000)
        else:
            big += big[-rnd.randrange(1, min(len(big), 32768) + 1) :][: rnd.randrange(1, 3000)] if big else b"x"
    big = bytes(big)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    comp = co.compress(big) + co.flush()
    dec = compu.decoder_interface.zlib_hip(compu.ZlibMode.Gzip)
    out = bytearray()
    buf = bytearray(1 << 20)
    t0 = time.perf_counter()
    pos = 0
    calls = 0
    while True:
        chunk = comp[pos : pos + 8192]
        r = dec.decode(chunk, buf)
        calls += 1
        assert r.is_ok()
        out += buf[: len(buf) - r.output_remain]
        pos += len(chunk) - r.input_remain
        if r.status == compu.DecodeStatus.Finished:
            break
        if r.status == compu.DecodeStatus.NeedInput:
            assert pos < len(comp)
    dt = time.perf_counter() - t0
    assert bytes(out) == big and pos == len(comp)
    assert calls >= len(comp) // 8192
    assert dt < 60, f"{calls} calls took {dt:.1f} s: the stream is being decoded from its start again"
    # a corrupted trailer is still caught when the stream arrives in pieces
    dec.reset()
    bad = bytearray(comp)
    bad[-6] ^= 0x10
    pos = 0
    status = None
    while pos < len(bad):
        r = dec.decode(bytes(bad[pos : pos + 65536]), buf)
        if not r.is_ok():
            status = r.status
            break
        pos += min(65536, len(bad) - pos) - r.input_remain
        if r.status == compu.DecodeStatus.Finished:
            status = r.status
            break
    assert status == compu.DecodeError(-3)


def test_streaming_memory_stays_bounded_over_a_long_gzip(gpu, alice):
    """A 192 MiB gzip fed in 64 KiB pieces (compu's decode loop, src/decoder/mod.rs:323-335): the decoder drops input in
    front of the last block boundary and output it has handed on (keeping the 32 KiB window), and carries the trailer
    CRC as a running value -- its buffers stay O(window + piece) instead of O(stream).  zlib-ng's state is ~40 KiB
    (src/decoder/zlib_ng.rs:29-55)."""
    import compu_amd

    rnd = random.Random(77)
    total = 192 << 20
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    comp = bytearray()
    crc = 0
    made = 0
    while made < total:
        # text (matches far and near), noise (stored blocks) and runs (long self-overlapping matches), 1 MiB at a time
        kind = rnd.randrange(3)
        if kind == 0:
            s0 = rnd.randrange(0, len(alice) - 65536)
            blk = (alice[s0 : s0 + 65536]) * 16
        elif kind == 1:
            blk = rnd.randbytes(1 << 20)
        else:
            blk = bytes([rnd.randrange(256)]) * (1 << 20)
        crc = zlib.crc32(blk, crc)
        comp += co.compress(blk)
        made += len(blk)
    comp += co.flush()
    comp = bytes(comp)
    dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
    out = bytearray(256 << 10)
    got_crc, got_len, pos, peak_pin, peak_dev = 0, 0, 0, 0, 0
    free0 = gpu.cuda.mem_get_info()[0]
    min_free = free0
    status = None
    piece = 64 << 10
    calls = 0
    while status != compu_amd.DecodeStatus.Finished:
        chunk = comp[pos : pos + piece]
        r = dec.decode(chunk, out)
        assert r.is_ok(), (pos, r.status)
        n = len(out) - r.output_remain
        got_crc = zlib.crc32(memoryview(out)[:n], got_crc)
        got_len += n
        pos += len(chunk) - r.input_remain
        status = r.status
        calls += 1
        if calls % 64 == 0:
            pin, devb = dec.footprint()
            peak_pin, peak_dev = max(peak_pin, pin), max(peak_dev, devb)
            min_free = min(min_free, gpu.cuda.mem_get_info()[0])
        assert calls < 200000
    assert got_len == made and got_crc == crc and pos == len(comp)
    # one deflate block of level-1 output is at most a few hundred KiB compressed / ~1 MiB decoded: window + piece + block
    assert peak_pin <= 4 << 20 and peak_dev <= 24 << 20, (peak_pin, peak_dev)
    assert free0 - min_free <= 64 << 20, (free0, min_free)  # hipMemGetInfo: no growth with the stream's length


def test_streaming_input_limit_is_an_error_not_a_hang(gpu):
    """More than 256 MiB of input buffered behind one block boundary cannot be addressed by the kernels' 32-bit bit
    offsets: chip_decode answers Err(-4) (Z_MEM_ERROR) instead of wrapping around and asking for input for ever."""
    import compu_amd

    dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
    hdr = bytes([0x1F, 0x8B, 8, 8, 0, 0, 0, 0, 0, 3])  # FNAME set: the header ends at a zero byte that never comes
    out = bytearray(1024)
    r = dec.decode(hdr + b"a" * 4096, out)
    assert r.status == compu_amd.DecodeStatus.NeedInput
    big = np.full((256 << 20) + 1, 0x61, np.uint8)
    r = dec.decode(big, out)
    assert not r.is_ok() and r.status.as_raw() == -4
    assert dec.reset()
    import zlib as _z
    co = _z.compressobj(6, _z.DEFLATED, 31)
    small = co.compress(b"hello") + co.flush()
    r = dec.decode(small, out)
    assert r.status == compu_amd.DecodeStatus.Finished and bytes(out[:5]) == b"hello"


def test_allocator_hooks_are_balanced_and_per_object(gpu):
    """chip_set_allocator (src/mem.rs:52-57,74-76 routing): host state of decoders and encoders comes from the hooks that
    were installed when the object was made and goes back to the SAME hooks, whatever is installed later."""
    import ctypes as C

    import compu_amd

    L = compu_amd.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    MALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)

    def make():
        live, stats = set(), {"malloc": 0, "free": 0, "foreign": 0}

        def m(opaque, n):
            p = libc.malloc(n)
            live.add(p)
            stats["malloc"] += 1
            return p

        def f(opaque, p):
            if p in live:
                live.remove(p)
            else:
                stats["foreign"] += 1
            stats["free"] += 1
            libc.free(p)

        return MALLOC(m), FREE(f), live, stats

    mA, fA, liveA, stA = make()
    mB, fB, liveB, stB = make()
    try:
        L.chip_set_allocator(mA, fA, None)
        dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
        enc = compu_amd.encoder_interface.zlib_hip(compu_amd.ZlibOptions().mode(compu_amd.ZlibMode.Gzip).compression(1))
        assert stA["malloc"] == 2 and len(liveA) == 2
        L.chip_set_allocator(mB, fB, None)  # later objects use B; the ones above still belong to A
        dec2 = compu_amd.decoder_interface.zstd_hip(compu_amd.ZstdOptions())
        assert stB["malloc"] == 1
        assert dec.reset() and enc.reset() and dec2.reset()
        out = bytearray(64)
        r = enc.encode(b"abc", out, compu_amd.EncodeOp.Finish)
        n = len(out) - r.output_remain
        back = bytearray(16)
        r2 = dec.decode(bytes(out[:n]), back)
        assert r2.status == compu_amd.DecodeStatus.Finished and bytes(back[:3]) == b"abc"
        del dec, enc, dec2
        import gc

        gc.collect()
        assert stA == {"malloc": 2, "free": 2, "foreign": 0} and not liveA
        assert stB == {"malloc": 1, "free": 1, "foreign": 0} and not liveB
    finally:
        L.chip_set_allocator(None, None, None)


def test_pinned_and_device_buffer_types(gpu, alice):
    """The north star's buffer types (src/buffer.rs grows pinned-host + device buffers): PinnedBuffer behaves like Buffer<N>
    (tests/decoder.rs:46-58 replayed over it); DeviceBuffer takes a batch from device memory to device memory."""
    import compu_amd

    comp = golden("alice29.txt.compressed.gz")
    dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
    buf = compu_amd.PinnedBuffer(4096)
    rest, out = comp, bytearray()
    while True:
        used, status = buf.decode(dec, rest)
        rest = rest[used:]
        out += buf.data()
        buf.consume()
        if status == compu_amd.DecodeStatus.Finished:
            break
    assert bytes(out) == alice
    enc = compu_amd.encoder_interface.zlib_hip(compu_amd.ZlibOptions().mode(compu_amd.ZlibMode.Gzip).compression(1))
    rest, packed = alice, bytearray()
    while True:
        used, status = buf.encode(enc, rest, compu_amd.EncodeOp.Finish)
        rest = rest[used:]
        packed += buf.data()
        buf.consume()
        assert status != compu_amd.EncodeStatus.Error
        if status == compu_amd.EncodeStatus.Finished:
            break
    assert zlib.decompress(bytes(packed), 31) == alice
    buf.close()
    # device to device: three units behind bytes the buffer already holds
    n = 3
    stride_in, stride_out = (len(comp) + 3) & ~3, (len(alice) + 15) & ~15
    src = compu_amd.DeviceBuffer(n * stride_in)
    for _ in range(n):
        src.upload(comp + b"\0" * (stride_in - len(comp)))
    dst = compu_amd.DeviceBuffer(16 + n * stride_out)
    dst.upload(bytes(range(16)))
    ol, iu, st = dst.decode_batch(31, src, [i * stride_in for i in range(n)], [len(comp)] * n, [i * stride_out for i in range(n)], [len(alice)] * n,
                                  n * stride_out)
    assert (st == 2).all() and (ol == len(alice)).all() and (iu == len(comp)).all()
    assert len(dst) == 16 + n * stride_out and dst.spare_capacity_len() == 0
    back = dst.download()
    assert bytes(back[:16]) == bytes(range(16))
    for i in range(n):
        assert bytes(back[16 + i * stride_out : 16 + i * stride_out + len(alice)]) == alice
    dst.consume()
    assert len(dst) == 0
