"""The reference's encoder tests (tests/encoder.rs:10-78 test_case, :115-173 test_case_empty_final; zlib-ng
variants :216-225 gzip, :249-258 zlib, :282-291 deflate, :345-354/:378-387/:411-420 empty-final) replayed
against the hip encoder + hip decoder, and batch parity of the GPU encoder with the oracle's bytes."""
import os
import random
import zlib

import numpy as np
import pytest

from conftest import golden
from test_inflate_gpu import _mk

pytestmark = pytest.mark.gpu
DATA = ["10x10y", "alice29.txt"]


def _test_case(c, encoder, decoder, data, expected_detection):
    """tests/encoder.rs:10-78"""
    EncodeOp, EncodeStatus, DecodeStatus = c.EncodeOp, c.EncodeStatus, c.DecodeStatus
    compressed = c.Vec(len(data))
    compressed.set_len(len(data))
    result = encoder.encode(data, compressed._buf, EncodeOp.Finish, 0, len(data))  # :17-18
    assert result.input_remain == 0
    if result.status == EncodeStatus.NeedOutput:  # :20-30
        compressed.reserve(100)
        spare = compressed.spare_capacity_len()
        result = encoder.encode(b"", compressed._buf, EncodeOp.Finish, len(compressed), spare)
        assert result.status == EncodeStatus.Finished
        compressed.set_len(len(compressed) + spare - result.output_remain)
    else:
        compressed.truncate(len(compressed) - result.output_remain)  # :32
    comp = bytes(compressed)
    assert c.Detection.detect(comp) == expected_detection  # :35
    out = bytearray(len(data))
    result = decoder.decode(comp, out)  # :36-38
    assert result.status == DecodeStatus.Finished and bytes(out) == data

    # Buffered encoder (:41-57)
    encoder.reset()
    buffer = c.Buffer(4096)
    buffer_input, compressed_full = data, bytearray()
    while True:
        consumed, status = buffer.encode(encoder, buffer_input, EncodeOp.Finish)
        buffer_input = buffer_input[consumed:]
        compressed_full += buffer.data()
        buffer.consume()
        assert status != EncodeStatus.Error
        if status == EncodeStatus.Finished:
            break
    assert bytes(compressed_full) == comp

    # Full vec encoding (:60-66)
    encoder.reset()
    vec = c.Vec()
    result = encoder.encode_vec_full(data, vec, EncodeOp.Finish)
    assert result.status == EncodeStatus.Finished and result.input_remain == 0
    assert bytes(vec) == comp

    decoder.reset()
    dvec = c.Vec()
    result = decoder.decode_vec_full(bytes(vec), dvec)  # :68-74
    assert result.status == DecodeStatus.Finished and bytes(dvec) == data
    encoder.reset()
    decoder.reset()
    return comp


def _test_case_empty_final(c, encoder, decoder, data):
    """tests/encoder.rs:115-173"""
    EncodeOp, EncodeStatus, DecodeStatus = c.EncodeOp, c.EncodeStatus, c.DecodeStatus
    compressed = c.Vec(len(data))
    result = encoder.encode(data, compressed._buf, EncodeOp.Process, 0, compressed.spare_capacity_len())  # :124-129
    assert result.status != EncodeStatus.Error
    compressed.set_len(compressed.spare_capacity_len() - result.output_remain)
    spare = compressed.spare_capacity_len()
    result = encoder.encode(data[len(data) - result.input_remain :], compressed._buf, EncodeOp.Flush, len(compressed), spare)  # :131-138
    assert result.input_remain == 0
    assert result.status == EncodeStatus.Continue
    compressed.set_len(len(compressed) + spare - result.output_remain)
    compressed.reserve(100)  # :140-148
    spare = compressed.spare_capacity_len()
    result = encoder.encode(b"", compressed._buf, EncodeOp.Finish, len(compressed), spare)
    if result.status == EncodeStatus.NeedOutput:
        # the reference can count on zlib-ng's compression ratio for 100 spare bytes; this level-1 class encoder
        # may need more room for the same flush -- keep giving room the way encode_vec_full does
        compressed.set_len(len(compressed) + spare - result.output_remain)
        vec_rest = c.Vec()
        result = encoder.encode_vec_full(b"", vec_rest, EncodeOp.Finish)
        compressed.extend_from_slice(bytes(vec_rest))
    else:
        compressed.set_len(len(compressed) + spare - result.output_remain)
    assert result.status == EncodeStatus.Finished
    comp = bytes(compressed)
    decompressed = bytearray(len(data) + 100)
    got = 0
    step = max(1, len(comp) // 4)
    for off in range(0, len(comp), step):  # :151-170
        chunk = comp[off : off + step]
        result = decoder.decode(chunk, decompressed, got, len(decompressed) - got)
        assert result.input_remain == 0
        assert result.output_remain > 0
        got = len(decompressed) - result.output_remain
        assert result.is_ok()
        if result.status == DecodeStatus.Finished:
            break
        assert result.status == DecodeStatus.NeedInput
    assert bytes(decompressed[:got]) == data
    encoder.reset()
    decoder.reset()


@pytest.mark.parametrize("mode_name,detection", [("Gzip", "Gzip"), ("Zlib", "Zlib"), ("Deflate", "Unknown")])
def test_should_encode_and_decode_zlib_hip(gpu, mode_name, detection):
    import compu_amd as c

    mode = getattr(c.ZlibMode, mode_name)
    for level in (1, 9, 0):
        encoder = c.encoder_interface.zlib_hip(c.ZlibOptions().mode(mode).compression(level))
        decoder = c.decoder_interface.zlib_hip(mode)
        assert encoder is not None and decoder is not None
        for name in DATA:
            comp = _test_case(c, encoder, decoder, golden(name), getattr(c.Detection, detection))
            wb = {"Gzip": 31, "Zlib": 15, "Deflate": -15}[mode_name]
            assert zlib.decompress(comp, wb) == golden(name)  # any conformant inflater takes the stream
        for name in DATA:
            _test_case_empty_final(c, encoder, decoder, golden(name))


def test_batch_encode_matches_oracle_bytes(gpu, alice):
    import compu_amd as c
    from oracle import oracle as O

    torch = gpu
    rnd = random.Random(31)
    for fmt in (-15, 15, 31):
        datas = []
        for it in range(120):
            # (sizes past 64 KiB: the hash table's 16-bit positions wrap; far copies alias positions 64 KiB apart)
            n = rnd.choice([0, 1, 3, 4, 5, 63, 64, 65, 100, 1000, 5000, 65535, 65536, 65537, 70000, 131071, 200000, 400000])
            kind = rnd.randrange(7)
            if kind < 5:
                datas.append(_mk(kind, n, rnd, alice))
            else:  # a block repeated with a period of (or just off) 64 KiB: the slot's low 16 bits name the copy one period back
                period = 65536 if kind == 5 else rnd.choice([65535, 65537, 70000, 32768, 32769])
                blk = (alice * 2)[: period // 2] + rnd.randbytes(period - period // 2)
                datas.append((blk * (n // period + 1))[:n])
        level = {-15: 1, 15: 0, 31: 1}[fmt]
        lens = np.array([len(d) for d in datas], np.int32)
        offs = np.zeros(len(datas), np.int64)
        offs[1:] = np.cumsum(((lens[:-1].astype(np.int64) + 3) & ~3) + rnd.choice([0, 1, 2, 3]))
        buf = np.zeros(int(offs[-1] + lens[-1]) + 8, np.uint8)
        for d, o in zip(datas, offs):
            buf[o : o + len(d)] = np.frombuffer(d, np.uint8)
        caps = np.array([c.encode_bound(fmt, len(d)) for d in datas], np.int32)
        ooff = np.zeros(len(datas), np.int64)
        ooff[1:] = np.cumsum(caps[:-1].astype(np.int64) + 7)
        dev = "cuda:0"
        d_out = torch.full((int(ooff[-1] + caps[-1]) + 8,), 0xA5, dtype=torch.uint8, device=dev)
        out_len, status = c.encode_batch(fmt, level, torch.from_numpy(buf[: (len(buf) // 4) * 4]).to(dev), torch.from_numpy(offs).to(dev),
                                         torch.from_numpy(lens).to(dev), d_out, torch.from_numpy(ooff).to(dev), torch.from_numpy(caps).to(dev))
        torch.cuda.synchronize()
        h = d_out.cpu().numpy()
        ol, st = out_len.cpu().numpy(), status.cpu().numpy()
        for i, d in enumerate(datas):
            assert st[i] == 2, (fmt, i, st[i])
            comp = bytes(h[ooff[i] : ooff[i] + ol[i]])
            # bytes between out_len and out_cap are scratch (a discarded fixed-Huffman attempt may sit there);
            # nothing beyond the capacity may be touched
            assert (h[ooff[i] + caps[i] : ooff[i] + caps[i] + 7] == 0xA5).all()
            e = O.DeflateEncoder(fmt, level)
            ref, ir, orr, est = e.encode(d, int(caps[i]) + 64, O.OP_FINISH)
            assert est == O.ENC_FINISHED
            assert comp == ref, (fmt, level, i, len(d), len(comp), len(ref))
            assert zlib.decompress(comp, fmt) == d


def test_dynamic_levels_match_oracle_bytes(gpu, alice):
    """Levels 2..9 write dynamic-Huffman blocks (oracle_deflate.c write_block): units of every kind, sizes around the
    chunk and the 65 536-token block limits (several blocks per unit, stored and fixed blocks among them), all strategies."""
    import compu_amd as c
    from oracle import oracle as O

    torch = gpu
    rnd = random.Random(77)
    dev = "cuda:0"
    for fmt, strategy in ((-15, 0), (15, 0), (31, 0), (-15, 2), (15, 3), (31, 1), (-15, 4)):
        datas = []
        for it in range(90):
            n = rnd.choice([0, 1, 3, 4, 5, 63, 64, 65, 100, 1000, 5000, 16384, 40000, 65471, 65472, 65473, 65535, 65536, 70000, 200000])
            datas.append(_mk(rnd.randrange(5), n, rnd, alice))
        datas.append((alice * 6)[:900000])
        datas.append(os.urandom(40000) + alice[:60000] + bytes(50000) + os.urandom(30000))
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
                                         torch.from_numpy(lens).to(dev), d_out, torch.from_numpy(ooff).to(dev), torch.from_numpy(caps).to(dev),
                                         strategy=strategy)
        torch.cuda.synchronize()
        h = d_out.cpu().numpy()
        ol, st = out_len.cpu().numpy(), status.cpu().numpy()
        kinds = set()
        for i, d in enumerate(datas):
            assert st[i] == 2, (fmt, i, st[i])
            comp = bytes(h[ooff[i] : ooff[i] + ol[i]])
            assert (h[ooff[i] + caps[i] : ooff[i] + caps[i] + 7] == 0xA5).all()
            e = O.DeflateEncoder(fmt, level, strategy)
            ref, ir, orr, est = e.encode(d, int(caps[i]) + 64, O.OP_FINISH)
            assert est == O.ENC_FINISHED
            assert comp == ref, (fmt, level, strategy, i, len(d), len(comp), len(ref))
            assert zlib.decompress(comp, fmt) == d
            body = comp[{-15: 0, 15: 2, 31: 10}[fmt]:]
            if body:
                kinds.add((body[0] >> 1) & 3)
        assert kinds >= ({1} if strategy == 4 else {0, 1, 2}), kinds  # stored, fixed and dynamic first blocks all occurred


def test_ratio_against_zlib(gpu, alice):
    """Stated bounds on LZ-compressible input: the default level (dynamic Huffman) stays within 1.15x of zlib level 1,
    level 1 (fixed Huffman, BASELINE.json cfg3) within 1.5x."""
    import compu_amd as c
    from bench_support import synth

    for level, bound in ((-1, 1.15), (1, 1.5)):
        enc = c.encoder_interface.zlib_hip(c.ZlibOptions().mode(c.ZlibMode.Deflate).compression(level))
        for data in (alice, synth.payloads(4).tobytes()):
            vec = c.Vec()
            r = enc.encode_vec_full(data, vec, c.EncodeOp.Finish)
            assert r.status == c.EncodeStatus.Finished
            assert zlib.decompress(bytes(vec), -15) == data
            assert len(vec) <= bound * len(zlib.compress(data, 1)), (level, len(vec), len(zlib.compress(data, 1)))
            enc.reset()


def test_host_memory_encode_matches_the_oracle(gpu):
    """chip_encode_batch_host (units in host memory, sliced over two streams): the same bytes as the oracle encoder."""
    import random
    import zlib

    import numpy as np

    import compu_amd
    from oracle import oracle as O

    rnd = random.Random(21)
    alice = golden("alice29.txt")
    datas = []
    for it in range(120):
        n = rnd.choice([0, 1, 100, 4000, 65536, 90000])
        k = rnd.randrange(3)
        if k == 0:
            s0 = rnd.randrange(0, max(1, len(alice) - n))
            datas.append(alice[s0 : s0 + n])
        elif k == 1:
            datas.append(rnd.randbytes(n))
        else:
            datas.append(bytes(rnd.choice(b"abcd") for _ in range(n)))
    lens = np.array([len(d) for d in datas], np.uint32)
    offs = np.zeros(len(datas), np.uint64)
    offs[1:] = np.cumsum(lens[:-1].astype(np.uint64))
    buf = np.zeros(int(lens.astype(np.uint64).sum()) + 8, np.uint8)
    buf[: int(lens.astype(np.uint64).sum())] = np.frombuffer(b"".join(datas), np.uint8)
    caps = np.array([compu_amd.encode_bound(31, len(d)) for d in datas], np.uint32)
    ooff = np.zeros(len(datas), np.uint64)
    ooff[1:] = np.cumsum((caps[:-1].astype(np.uint64) + 15) & ~np.uint64(15))
    out = np.zeros(int(ooff[-1] + caps[-1]) + 16, np.uint8)
    ol, st = compu_amd.encode_batch_host(31, 1, buf, offs, lens, out, ooff, caps, slice_bytes=1 << 19)
    for i, d in enumerate(datas):
        got = bytes(out[int(ooff[i]) : int(ooff[i]) + int(ol[i])])
        assert st[i] == 2, (i, st[i])
        assert zlib.decompress(got, 31) == d, i
        enc = O.DeflateEncoder(O.MODE_GZIP, 1)
        ref, in_rem, out_rem, est = enc.encode(d, int(caps[i]), O.OP_FINISH)
        assert got == ref, i


def test_zlib_options_surface_strategy_mem_level_default_level(gpu, alice):
    """Every field of ZlibOptions (src/encoder/zlib_common.rs:47-103) reaches the backend as deflateInit2_ would get it
    (src/encoder/zlib_ng.rs:69-79): strategy (HuffmanOnly: no matches, Rle: distance 1 only), mem_level 1..9,
    compression -1 = zlib's default; what zlib refuses (Z_STREAM_ERROR -> None) is refused.  GPU bytes == oracle bytes."""
    import compu_amd
    from compu_amd import ZlibMode, ZlibOptions, ZlibStrategy, EncodeOp, EncodeStatus
    from oracle import oracle as O

    runs = bytes([7]) * 5000 + alice[:20000] + b"ab" * 3000 + bytes(range(256)) * 8
    for strat in ZlibStrategy:
        for level in (-1, 1, 9):
            enc = compu_amd.encoder_interface.zlib_hip(ZlibOptions().mode(ZlibMode.Zlib).compression(level).strategy(strat).mem_level(9 if level == 9 else 1))
            assert enc is not None
            out = compu_amd.Vec(0)
            r = enc.encode_vec_full(runs, out, EncodeOp.Finish)
            assert r.status == EncodeStatus.Finished
            comp = bytes(out)
            assert zlib.decompress(comp) == runs
            ref = O.DeflateEncoder(O.MODE_ZLIB, 6 if level == -1 else level, int(strat))
            want, _ir, _or, st = ref.encode(runs, len(runs) + 1024, 2)
            assert st == 2 and comp == want, (strat, level)
        # what the strategy means, seen from the decoder: tokens of the fixed-Huffman stream
    sizes = {}
    for strat in ZlibStrategy:
        enc = compu_amd.encoder_interface.zlib_hip(ZlibOptions().mode(ZlibMode.Deflate).compression(1).strategy(strat))
        out = compu_amd.Vec(0)
        enc.encode_vec_full(runs, out, EncodeOp.Finish)
        sizes[strat] = len(bytes(out))
    assert sizes[ZlibStrategy.Default] == sizes[ZlibStrategy.Filtered] == sizes[ZlibStrategy.Fixed]
    assert sizes[ZlibStrategy.HuffmanOnly] > sizes[ZlibStrategy.Rle] > sizes[ZlibStrategy.Default]  # no matches > runs only > all matches
    # refused like deflateInit2_ refuses them
    L = compu_amd.lib()
    from compu_amd.api import _EncoderOpts
    import ctypes as C

    for bad in (_EncoderOpts(31, 1, -1, 5, 8), _EncoderOpts(31, 1, -1, 0, 10), _EncoderOpts(31, 10, -1, 0, 8), _EncoderOpts(31, -2, -1, 0, 8), _EncoderOpts(47, 1, -1, 0, 8)):
        assert not L.chip_encoder_new(C.byref(bad))


def test_committed_encoder_digests(gpu):
    """The kernels reproduce tests/golden/encoder_digests.json (what oracle_deflate.c's algorithm gives for the reference's
    two fixtures, tools/make_encoder_digests.py): the pair oracle + kernels cannot drift together unnoticed."""
    import hashlib
    import json

    import compu_amd as c
    from conftest import GOLDEN

    want = json.load(open(os.path.join(GOLDEN, "encoder_digests.json")))
    for key, w in want.items():
        name, mode, level, strategy = key.split(":")
        opts = c.ZlibOptions().mode(c.ZlibMode.Gzip).compression(int(level[5:])).strategy(c.ZlibStrategy(int(strategy[8:])))
        enc = c.encoder_interface.zlib_hip(opts)
        vec = c.Vec()
        r = enc.encode_vec_full(golden(name), vec, c.EncodeOp.Finish)
        assert r.status == c.EncodeStatus.Finished
        comp = bytes(vec)
        assert (len(comp), hashlib.sha256(comp).hexdigest()) == (w["size"], w["sha256"]), key
