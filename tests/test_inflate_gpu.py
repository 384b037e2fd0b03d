"""Parity of the HIP inflate path (through the C ABI) against the oracle and the original bytes."""
import os
import random
import zlib

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _pack(parts):
    lens = np.array([len(p) for p in parts], dtype=np.int32)
    offs = np.zeros(len(parts), dtype=np.int64)
    if len(parts) > 1:
        offs[1:] = np.cumsum(lens[:-1].astype(np.int64))
    total = int(lens.astype(np.int64).sum())
    buf = np.zeros(((total + 3) & ~3) + 4, np.uint8)
    buf[:total] = np.frombuffer(b"".join(parts), dtype=np.uint8)
    return buf, offs, lens


def run_batch(torch, fmt, parts, caps, check_tail=True, flags=0):
    """Decode `parts` in one launch; returns (list of outputs, out_len, in_used, status) on the host."""
    import compu_amd

    buf, offs, lens = _pack(parts)
    caps = np.asarray(caps, dtype=np.int32)
    ooff = np.zeros(len(parts), dtype=np.int64)
    if len(parts) > 1:
        ooff[1:] = np.cumsum(((caps[:-1].astype(np.int64) + 15) & ~15))
    total_out = int(ooff[-1] + caps[-1]) + 16
    dev = "cuda:0"
    d_in = torch.from_numpy(buf).to(dev)
    d_out = torch.full((total_out,), 0xA5, dtype=torch.uint8, device=dev)
    out_len, in_used, status = compu_amd.decode_batch(
        fmt, d_in, torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev), d_out,
        torch.from_numpy(ooff).to(dev), torch.from_numpy(caps).to(dev), flags=flags)
    torch.cuda.synchronize()
    h_out = d_out.cpu().numpy()
    ol, iu, st = out_len.cpu().numpy(), in_used.cpu().numpy(), status.cpu().numpy()
    outs = [bytes(h_out[ooff[i] : ooff[i] + ol[i]]) for i in range(len(parts))]
    # bytes behind each unit's produced range must be untouched (poison intact)
    for i in range(len(parts)):
        # zstd parks regenerated literals at the end of the unit's capacity (documented scratch use)
        lo = ooff[i] + (ol[i] if check_tail else caps[i])
        tail = h_out[lo : ooff[i] + ((caps[i] + 15) & ~15)]
        assert (tail == 0xA5).all(), f"unit {i}: wrote past its output range"
    return outs, ol, iu, st


def oracle_batch(mode, parts, caps):
    from oracle import oracle as O

    res = []
    for p, c in zip(parts, caps):
        d = O.InflateDecoder(mode)
        got, ir, orr, st, err = d.decode(p, int(c))
        res.append((got, len(p) - ir, err if err else st))
    return res


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


def test_raw_deflate_small_units(gpu, alice):
    rnd = random.Random(11)
    datas, parts = [], []
    for it in range(300):
        n = rnd.choice([0, 1, 2, 3, 10, 100, 1000, 5000, 20000, 65536, 70000])
        data = _mk(rnd.randrange(5), n, rnd, alice)
        level = rnd.choice([0, 1, 4, 6, 9])
        strat = rnd.choice([0, 0, 0, 1, 2, 3, 4])
        co = zlib.compressobj(level, zlib.DEFLATED, -15, rnd.choice([1, 8, 9]), strat)
        comp = co.compress(data[: n // 2]) + (co.flush(zlib.Z_SYNC_FLUSH) if rnd.random() < 0.3 else b"") + co.compress(data[n // 2 :]) + co.flush()
        datas.append(data)
        parts.append(comp)
    caps = [len(d) + rnd.choice([0, 0, 7, 100]) for d in datas]
    outs, ol, iu, st = run_batch(gpu, -15, parts, caps)
    ref = oracle_batch(-15, parts, caps)
    for i in range(len(parts)):
        assert st[i] == 2, (i, st[i])
        assert outs[i] == datas[i], i
        assert outs[i] == ref[i][0] and iu[i] == ref[i][1] and st[i] == ref[i][2], i


@pytest.mark.timeout(1500)
def test_full_size_batches_match_the_oracle(gpu):
    """BASELINE.json's full size -- 65,536 units x 64 KiB -- for configs[1] (stored, fixed Huffman), the dynamic-Huffman config
    and the mixed gzip+zstd config:
    the GPU output equals the payload on the device, the oracle's output of the same units equals the payload on the host
    (so GPU == oracle, byte for byte, without moving 4 GiB twice), and out_len / in_used / status equal the oracle's."""
    import compu_amd
    from bench_support import synth
    from oracle import oracle as O

    n = 65536
    threads = min(32, len(os.sched_getaffinity(0)))
    pay = synth.payloads(n, threads=threads)
    dev = "cuda:0"
    d_pay = gpu.from_numpy(pay).to(dev)
    ooff = np.arange(n, dtype=np.uint64) * synth.UNIT
    caps = np.full(n, synth.UNIT, np.uint32)
    for kind in ("stored", "fixed", "dynamic", "mixed"):
        if kind == "mixed":
            packed, offs, lens = synth.mixed_units(pay, n, threads=threads)
        else:
            packed, offs, lens = synth.deflate_units(pay, n, kind=kind, threads=threads)
        d_out = gpu.zeros(n * synth.UNIT, dtype=gpu.uint8, device=dev)
        ol, iu, st = compu_amd.decode_batch(
            0 if kind == "mixed" else -15, gpu.from_numpy(packed).to(dev), gpu.from_numpy(offs.astype(np.int64)).to(dev),
            gpu.from_numpy(lens.astype(np.int32)).to(dev), d_out, gpu.from_numpy(ooff.astype(np.int64)).to(dev), gpu.from_numpy(caps.astype(np.int32)).to(dev))
        gpu.cuda.synchronize()
        assert gpu.equal(d_out, d_pay), kind
        del d_out
        ol, iu, st = ol.cpu().numpy().astype(np.uint32), iu.cpu().numpy().astype(np.uint32), st.cpu().numpy()
        if kind == "mixed":
            is_gz = packed[offs.astype(np.int64)] == 0x1F
            gz, zs = np.flatnonzero(is_gz), np.flatnonzero(~is_gz)
            assert len(gz) > n // 3 and len(zs) > n // 3
            ref = np.zeros(n * synth.UNIT, np.uint8)  # both oracles decode into the one buffer, each its own units
            r_len, r_st = np.zeros(n, np.uint32), np.zeros(n, np.int32)
            _o, r_len[gz], r_st[gz], bad_g = O.inflate_units(O.MODE_GZIP, packed, offs[gz], lens[gz], n * synth.UNIT, ooff[gz], caps[gz], threads=threads, out=ref)
            _o, r_len[zs], r_st[zs], bad_z = O.zstd_units(packed, offs[zs], lens[zs], n * synth.UNIT, ooff[zs], caps[zs], threads=threads, out=ref)
            assert bad_g == 0 and bad_z == 0 and np.array_equal(ref, pay)
            assert (st == r_st).all() and (ol == r_len).all() and (st == 2).all() and (iu == lens).all()
            del ref
        else:
            ref, r_len, r_st, bad = O.inflate_units(O.MODE_DEFLATE, packed, offs, lens, n * synth.UNIT, ooff, caps, threads=threads)
            assert bad == 0 and np.array_equal(ref, pay)
            assert (st == r_st).all() and (ol == r_len).all() and (st == 2).all() and (iu == lens).all()
            del ref
        del packed


def test_raw_deflate_64k_synthetic_units(gpu):
    from bench_support import synth

    n = 256
    pay = synth.payloads(n)
    for kind in ("stored", "fixed", "dynamic", "level1"):
        packed, offs, lens = synth.deflate_units(pay, n, kind=kind)
        parts = [bytes(packed[int(offs[i]) : int(offs[i]) + int(lens[i])]) for i in range(n)]
        outs, ol, iu, st = run_batch(gpu, -15, parts, [65536] * n)
        assert (st == 2).all(), (kind, st[st != 2][:8])
        assert (ol == 65536).all()
        assert (iu == lens).all()
        assert b"".join(outs) == pay.tobytes(), kind


def _skewed_payload(n_lit_syms, n, seed):
    """Bytes whose literal AND match-distance statistics follow the Fibonacci numbers (the counts that make a Huffman tree as deep
    as it can get): zlib's codes then reach its 15-bit limit on both alphabets within one 16 K-token block."""
    rnd = random.Random(seed)
    syms = list(range(256))
    rnd.shuffle(syms)
    syms = syms[:n_lit_syms]
    fib = [1, 1]
    while len(fib) < 64:
        fib.append(fib[-1] + fib[-2])
    lw = [fib[min(k, 40)] for k in range(n_lit_syms)]
    dcodes = list(range(30))
    rnd.shuffle(dcodes)
    dw = [fib[k] for k in range(30)]
    out = bytearray(bytes(rnd.choices(syms, weights=lw, k=64)))
    while len(out) < n:
        if rnd.random() < 0.5:
            out.append(rnd.choices(syms, weights=lw, k=1)[0])
            continue
        d_code = rnd.choices(dcodes, weights=dw, k=1)[0]
        if d_code < 4:
            dist = d_code + 1
        else:
            eb = (d_code >> 1) - 1
            dist = 1 + ((2 + (d_code & 1)) << eb) + rnd.randrange(1 << eb)
        if dist > len(out):
            continue
        for _ in range(rnd.choice([3, 3, 3, 4, 5, 9])):
            out.append(out[-dist])
    return bytes(out[:n])


def test_deep_huffman_codes_use_the_subtables(gpu):
    """Codes longer than the 9-bit (literal/length) and 8-bit (distance) roots: geometric symbol statistics drive zlib's trees to
    15 bits, so most lookups of these streams go through the second-level tables; the oracle decides what is right."""
    torch = gpu
    parts, caps, want = [], [], []
    for seed, (nsym, n) in enumerate([(40, 70000), (120, 200000), (256, 65536), (24, 300000), (200, 400000)]):
        pay = _skewed_payload(nsym, n, 100 + seed)
        for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_FILTERED), (1, zlib.Z_DEFAULT_STRATEGY)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
            parts.append(co.compress(pay) + co.flush())
            caps.append(len(pay))
            want.append(pay)
    outs, ol, iu, st = run_batch(torch, -15, parts, caps)
    ref = oracle_batch(-15, parts, caps)
    for i in range(len(parts)):
        assert st[i] == ref[i][2] == 2, (i, st[i], ref[i][2])
        assert outs[i] == want[i] == ref[i][0], f"stream {i} differs"
        assert iu[i] == len(parts[i]) == ref[i][1]


def test_truncated_corrupt_and_small_caps_match_oracle(gpu, alice):
    rnd = random.Random(5)
    parts, caps = [], []
    for it in range(400):
        n = rnd.choice([50, 500, 5000, 30000])
        data = _mk(rnd.choice([1, 2, 4]), n, rnd, alice)
        co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, 8, rnd.choice([0, 4]))
        comp = bytearray(co.compress(data) + co.flush())
        mode = rnd.randrange(4)
        if mode == 0:  # flip a bit
            k = rnd.randrange(len(comp))
            comp[k] ^= 1 << rnd.randrange(8)
        elif mode == 1:  # truncate
            comp = comp[: rnd.randrange(len(comp))]
        elif mode == 2:  # trailing garbage
            comp += rnd.randbytes(rnd.randrange(1, 9))
        cap = n if mode != 3 else rnd.randrange(0, n)  # mode 3: output too small
        parts.append(bytes(comp))
        caps.append(cap)
    outs, ol, iu, st = run_batch(gpu, -15, parts, caps)
    ref = oracle_batch(-15, parts, caps)
    for i in range(len(parts)):
        r_out, r_used, r_st = ref[i]
        assert outs[i] == r_out, (i, len(outs[i]), len(r_out))
        # The two places where the batch status is, by its documented contract (include/compu_hip.h, chip_decode_batch), the
        # cause rather than compu's mapping of zlib's return code -- each pinned to its exact condition, everything else equal:
        if r_st == 0 and len(r_out) == caps[i] and len(parts[i]) and r_used == len(parts[i]) and ol[i] == caps[i] and st[i] != 0:
            # output full AND every input byte consumed: zlib says Z_OK with avail_in == 0, which compu reads as NeedInput
            # (mod.rs:476-479); the batch API names the limit that was hit
            assert st[i] == 1, (i, st[i])
            continue
        if len(parts[i]) == 0:
            # empty input: zlib answers Z_BUF_ERROR (no progress), which compu maps to NeedOutput (mod.rs:481); the batch
            # API calls an empty unit what it is: truncated
            assert r_st == 1 and st[i] == 0 and ol[i] == 0, (i, st[i], r_st)
            continue
        assert st[i] == r_st, (i, st[i], r_st)
        if r_st == 2:
            assert iu[i] == r_used, (i, iu[i], r_used)


def test_compu_status_flag_reports_the_reference_mapping_verbatim(gpu, alice):
    """CHIP_F_COMPU_STATUS: the two pinned deviations above disappear -- status AND in_used are the oracle's (= zlib's
    avail_in at the moment the output filled, src/decoder/mod.rs:475-483) for every unit, whatever ran out."""
    import compu_amd

    rnd = random.Random(6)
    parts, caps = [], []
    for it in range(700):
        n = rnd.choice([50, 500, 5000, 30000, 120000])
        data = _mk(rnd.choice([1, 2, 4]), n, rnd, alice)
        co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, 8, rnd.choice([0, 4]))
        comp = bytearray(co.compress(data) + co.flush())
        mode = rnd.randrange(5)
        if mode == 0:
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            comp = comp[: rnd.randrange(len(comp))]
        cap = n if mode < 2 else rnd.randrange(0, n)  # modes 2..4: the output is too small
        if mode == 4:  # ... and the input ends close behind the token that does not fit
            ref = oracle_batch(-15, [bytes(comp)], [cap])[0]
            comp = comp[: ref[1] + rnd.choice([0, 0, 0, 1, 2])]
        parts.append(bytes(comp))
        caps.append(cap)
    parts += [b"", b""]
    caps += [0, 100]
    outs, ol, iu, st = run_batch(gpu, -15, parts, caps, flags=compu_amd.F_COMPU_STATUS)
    ref = oracle_batch(-15, parts, caps)
    flipped = 0
    for i in range(len(parts)):
        r_out, r_used, r_st = ref[i]
        assert outs[i] == r_out, (i, len(outs[i]), len(r_out))
        assert st[i] == r_st, (i, st[i], r_st, caps[i], len(parts[i]))
        if r_st in (1, 2) and len(parts[i]):
            assert iu[i] == r_used, (i, iu[i], r_used, r_st)
        flipped += r_st == 0 and len(r_out) == caps[i] and r_used == len(parts[i]) and len(parts[i]) > 0
    assert flipped >= 20  # the case the flag exists for was exercised
    # the flag changes nothing else: wrapped formats, finished units
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    whole = co.compress(alice[:70000]) + co.flush()
    gz = [whole, whole[:-3], b""]
    outs, ol, iu, st = run_batch(gpu, 31, gz, [70000, 70000, 10], flags=compu_amd.F_COMPU_STATUS)
    ref = oracle_batch(31, gz, [70000, 70000, 10])
    assert [int(x) for x in st] == [r[2] for r in ref] == [2, 0, 1]


def test_large_multiblock_streams(gpu, alice):
    """Streams far larger than a 64 KiB unit: many blocks of every type, window reloads, matches that reach the
    full 32 KiB window across block boundaries, highly compressible runs (258-byte matches, distance 1)."""
    import compu_amd

    rnd = random.Random(99)
    big = bytearray()
    while len(big) < 6_000_000:
        k = rnd.randrange(6)
        if k == 0:
            big += alice[rnd.randrange(len(alice) // 2) :][: rnd.randrange(1, 200000)]
        elif k == 1:
            big += rnd.randbytes(rnd.randrange(1, 70000))
        elif k == 2:
            big += b"\0" * rnd.randrange(1, 300000)
        elif k == 3:
            big += bytes(rnd.choice(b"abc") for _ in range(rnd.randrange(1, 20000)))
        elif k == 4:
            big += big[-rnd.randrange(1, min(len(big), 32768) + 1) :][: rnd.randrange(1, 5000)] if big else b"x"
        else:
            big += bytes([rnd.randrange(256)]) * rnd.randrange(1, 1000)
    big = bytes(big)
    cases = []
    for level, wb in ((6, 31), (1, -15), (9, 15), (0, -15)):
        co = zlib.compressobj(level, zlib.DEFLATED, wb)
        comp = b""
        pos = 0
        while pos < len(big):  # flushes force extra block boundaries, incl. empty stored blocks
            step = rnd.randrange(1, 400000)
            comp += co.compress(big[pos : pos + step])
            if rnd.random() < 0.3:
                comp += co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
            pos += step
        comp += co.flush()
        cases.append((wb, comp))
    # batch API
    for wb, comp in cases:
        outs, ol, iu, st = run_batch(gpu, wb, [comp], [len(big)])
        assert st[0] == 2 and iu[0] == len(comp) and outs[0] == big, (wb, st[0], ol[0])
    # streaming decoder in 64 KiB input chunks and 100 KB output buffers
    wb, comp = cases[0]
    dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
    out = bytearray()
    buf = bytearray(100_000)
    pos = 0
    while True:
        chunk = comp[pos : pos + 65536]
        r = dec.decode(chunk, buf)
        assert r.is_ok()
        out += buf[: len(buf) - r.output_remain]
        pos += len(chunk) - r.input_remain
        if r.status == compu_amd.DecodeStatus.Finished:
            break
    assert bytes(out) == big and pos == len(comp)


def test_trim_releases_and_reallocates_scratch(gpu):
    """chip_trim() frees the cached token scratch; the next launch allocates it again and decodes as before."""
    import compu_amd
    from bench_support import synth

    n = 64
    pay = synth.payloads(n)
    packed, offs, lens = synth.deflate_units(pay, n, kind="dynamic")
    parts = [bytes(packed[int(offs[i]) : int(offs[i]) + int(lens[i])]) for i in range(n)]
    for _ in range(2):
        outs, ol, iu, st = run_batch(gpu, -15, parts, [65536] * n)
        assert (st == 2).all() and b"".join(outs) == pay.tobytes()
        compu_amd.trim()


def test_host_memory_variant_matches_device_variant(gpu, alice):
    """chip_decode_batch_host: units in host memory, sliced over two streams; same results as the device call
    (several slices, units of every block type, a truncated and a corrupt unit, zstd and Detection routing)."""
    import compu_amd
    import zstd_ref

    rnd = random.Random(4)
    z = zstd_ref.load()
    datas, parts = [], []
    for it in range(200):
        n = rnd.choice([0, 100, 5000, 65536, 70000])
        data = _mk(rnd.randrange(5), n, rnd, alice)
        k = rnd.randrange(3)
        if k == 0:
            co = zlib.compressobj(rnd.choice([0, 1, 6]), zlib.DEFLATED, 31)
            comp = co.compress(data) + co.flush()
        elif k == 1:
            co = zlib.compressobj(6, zlib.DEFLATED, 15)
            comp = co.compress(data) + co.flush()
        else:
            comp = zstd_ref.compress(z, data, 3)
        if it == 17:
            comp = comp[: len(comp) // 2]
        if it == 33 and len(comp) > 20:
            comp = comp[:15] + bytes([comp[15] ^ 0x55]) + comp[16:]
        datas.append(data)
        parts.append(comp)
    caps = [len(d) + 64 for d in datas]
    outs_d, ol_d, iu_d, st_d = run_batch(gpu, 0, parts, caps, check_tail=False)  # CHIP_FMT_DETECT on the device path
    buf, offs, lens = _pack(parts)
    caps_a = np.asarray(caps, dtype=np.uint32)
    ooff = np.zeros(len(parts), dtype=np.uint64)
    ooff[1:] = np.cumsum(((caps_a[:-1].astype(np.uint64) + 15) & ~np.uint64(15)))
    out = np.zeros(int(ooff[-1] + caps_a[-1]) + 16, np.uint8)
    ol, iu, st = compu_amd.decode_batch_host(0, buf, offs.astype(np.uint64), lens.astype(np.uint32), out, ooff, caps_a, slice_bytes=1 << 20)
    assert (st == st_d).all() and (ol == ol_d).all() and (iu == iu_d).all()
    for i in range(len(parts)):
        assert bytes(out[int(ooff[i]) : int(ooff[i]) + int(ol[i])]) == outs_d[i], i
        if st[i] == 2:
            assert outs_d[i] == datas[i]


def test_host_batch_layouts_reverse_order_gaps_and_poison(gpu, alice):
    """chip_decode_batch_host with layouts the straight-through path cannot serve: output ranges in REVERSE unit order,
    gaps between units, short outputs.  Exactly out_len bytes are written per unit; every other byte of the caller's
    output keeps its poison (a slice used to be copied back as one span of uninitialised device memory)."""
    import compu_amd

    rnd = random.Random(9)
    datas, parts = [], []
    for it in range(120):
        n = rnd.choice([0, 10, 700, 5000, 40000, 65536])
        data = _mk(rnd.randrange(5), n, rnd, alice)
        co = zlib.compressobj(rnd.choice([1, 6]), zlib.DEFLATED, -15)
        datas.append(data)
        parts.append(co.compress(data) + co.flush())
    parts[7] = parts[7][: len(parts[7]) // 2]  # a truncated unit writes what it decoded and nothing more
    buf, offs, lens = _pack(parts)
    caps = np.array([len(d) + rnd.choice([0, 0, 33, 1000]) for d in datas], dtype=np.uint32)
    caps[11] = len(datas[11]) // 2  # too small: NeedOutput with exactly cap bytes
    gaps = np.array([rnd.choice([16, 48, 64, 1000]) for _ in datas], dtype=np.uint64)
    spans = ((caps.astype(np.uint64) + 15) & ~np.uint64(15)) + gaps
    for order in ("reverse", "forward"):
        ooff = np.zeros(len(parts), dtype=np.uint64)
        if order == "reverse":
            ooff[::-1][1:] = np.cumsum(spans[::-1][:-1])
        else:
            ooff[1:] = np.cumsum(spans[:-1])
        out = np.full(int(spans.sum()) + 64, 0xA5, np.uint8)
        ol, iu, st = compu_amd.decode_batch_host(-15, buf, offs.astype(np.uint64), lens.astype(np.uint32), out, ooff, caps, slice_bytes=300_000)
        ref = oracle_batch(-15, parts, caps)
        mask = np.ones(len(out), bool)
        for i in range(len(parts)):
            got = bytes(out[int(ooff[i]) : int(ooff[i]) + int(ol[i])])
            assert got == ref[i][0] and st[i] == ref[i][2], (order, i, st[i], ref[i][2])
            mask[int(ooff[i]) : int(ooff[i]) + int(ol[i])] = False
        assert (out[mask] == 0xA5).all(), f"{order}: bytes outside the produced ranges were written"
        assert st[7] == 0 and st[11] == 1 and ol[11] == caps[11]


def test_multi_device_entry_buckets_by_format(gpu, alice):
    """chip_decode_batch_multi: the host-side partition of SURVEY.md sec. 8e.  Two workers (both on device 0 here: a
    one-GPU box) take contiguous, byte-balanced unit ranges; a CHIP_FMT_DETECT batch is bucketed so that every launch is
    homogeneous (units interleave gzip / zlib / zstd / garbage / too-short).  Same results as the one-launch device path."""
    import compu_amd
    import zstd_ref

    rnd = random.Random(21)
    z = zstd_ref.load()
    datas, parts = [], []
    for it in range(300):
        n = rnd.choice([0, 300, 9000, 65536])
        data = _mk(rnd.randrange(5), n, rnd, alice)
        k = rnd.randrange(5)
        if k == 0:
            comp = zlib.compress(data, 6)
        elif k == 1:
            co = zlib.compressobj(6, zlib.DEFLATED, 31)
            comp = co.compress(data) + co.flush()
        elif k == 2:
            comp = zstd_ref.compress(z, data, 3)
        elif k == 3:
            comp = bytes(rnd.randrange(1, 255) for _ in range(rnd.randrange(4, 40)))  # Detection::Unknown (mostly)
        else:
            comp = b"\x1f"  # too short to classify
        datas.append(data)
        parts.append(comp)
    caps = [len(d) + 64 for d in datas]
    outs_d, ol_d, iu_d, st_d = run_batch(gpu, 0, parts, caps, check_tail=False)
    buf, offs, lens = _pack(parts)
    caps_a = np.asarray(caps, dtype=np.uint32)
    ooff = np.zeros(len(parts), dtype=np.uint64)
    ooff[1:] = np.cumsum(((caps_a[:-1].astype(np.uint64) + 15) & ~np.uint64(15)))
    for devices in ([0], [0, 0], None):
        out = np.full(int(ooff[-1] + caps_a[-1]) + 16, 0x5A, np.uint8)
        ol, iu, st = compu_amd.decode_batch_multi(0, buf, offs.astype(np.uint64), lens.astype(np.uint32), out, ooff, caps_a, devices=devices,
                                                  slice_bytes=1 << 20)
        assert (st == st_d).all() and (ol == ol_d).all() and (iu == iu_d).all(), devices
        for i in range(len(parts)):
            assert bytes(out[int(ooff[i]) : int(ooff[i]) + int(ol[i])]) == outs_d[i], (devices, i)
    assert set(np.unique(st_d).tolist()) >= {0, 2, 4}  # finished units, too-short (NeedInput) and Unknown ones all occur
    with pytest.raises(RuntimeError):
        compu_amd.decode_batch_multi(0, buf, offs, lens, out, ooff, caps_a, devices=[0, 99])  # no such device: refused, not ignored


def test_detect_batch_matches_host_detection(gpu):
    """chip_detect_batch (device) == chip_detect (host) == Detection::detect (src/decoder/mod.rs:28-114), every 2-byte
    prefix that can matter plus the zstd magic and short inputs."""
    import compu_amd

    parts = [b"", b"\x1f", b"\x1f\x8b", b"\x28\xb5\x2f", b"\x28\xb5\x2f\xfd", b"abcd", b"ab", b"\x28\xb5\x2f\xfd\x00\x00"]
    for cmf in (0x08, 0x18, 0x28, 0x38, 0x48, 0x58, 0x68, 0x78, 0x88):
        for flg in range(256):
            parts.append(bytes([cmf, flg]))
            parts.append(bytes([cmf, flg, 0, 0]))
    buf, offs, lens = _pack(parts)
    dev = "cuda:0"
    kind = compu_amd.detect_batch(gpu.from_numpy(buf).to(dev), gpu.from_numpy(offs).to(dev), gpu.from_numpy(lens).to(dev))
    gpu.cuda.synchronize()
    kind = kind.cpu().numpy()
    L = compu_amd.lib()
    for i, p in enumerate(parts):
        assert kind[i] == L.chip_detect(p, len(p)), (i, p, kind[i])
    with pytest.raises(TypeError):  # int32 offsets would be read as u64 on the device: refused on the host
        compu_amd.detect_batch(gpu.from_numpy(buf).to(dev), gpu.from_numpy(offs.astype(np.int32)).to(dev), gpu.from_numpy(lens).to(dev))


def test_concurrent_launches_from_two_host_threads(gpu):
    """Two host threads launch batches on the same (default) stream at the same time: the persistent kernel's unit
    counter is reset and consumed per launch, so neither batch may lose or repeat units."""
    import threading

    import compu_amd
    from bench_support import synth

    n = 512
    pay = synth.payloads(n)
    packed, offs, lens = synth.deflate_units(pay, n, kind="dynamic")
    dev = "cuda:0"
    d_in = gpu.from_numpy(packed).to(dev)
    d_off = gpu.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = gpu.from_numpy(lens.astype(np.int32)).to(dev)
    ooff = gpu.arange(n, dtype=gpu.int64, device=dev) * 65536
    caps = gpu.full((n,), 65536, dtype=gpu.int32, device=dev)
    outs = [gpu.zeros(n * 65536, dtype=gpu.uint8, device=dev) for _ in range(2)]
    res = [None, None]
    want = gpu.from_numpy(pay).to(dev)

    def work(k):
        for _ in range(20):
            res[k] = compu_amd.decode_batch(-15, d_in, d_off, d_len, outs[k], ooff, caps)

    # a third thread launches batches of growing size on the same stream: every growth frees and reallocates the cached token
    # scratch, which the other threads' launches use -- lookup, counter reset and launch are one critical section
    big_n = [64, 256, 1024, 2048, 4096]
    big_out = gpu.zeros(max(big_n) * 65536, dtype=gpu.uint8, device=dev)
    big_res = {}

    def grow():
        compu_amd.trim()  # start from no scratch at all
        for m in big_n:
            rep = (m + n - 1) // n
            o = d_off.repeat(rep)[:m].contiguous()
            l = d_len.repeat(rep)[:m].contiguous()
            oo = gpu.arange(m, dtype=gpu.int64, device=dev) * 65536
            cc = gpu.full((m,), 65536, dtype=gpu.int32, device=dev)
            big_res[m] = compu_amd.decode_batch(-15, d_in, o, l, big_out, oo, cc)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)] + [threading.Thread(target=grow)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    gpu.cuda.synchronize()
    for k in range(2):
        ol, iu, st = res[k]
        assert bool((st == 2).all()) and bool((ol == 65536).all())
        assert gpu.equal(outs[k], want)
    for m, (ol, iu, st) in big_res.items():
        assert bool((st == 2).all()) and bool((ol == 65536).all()), m
    m = big_n[-1]
    assert gpu.equal(big_out[: m * 65536].view(m // n, n * 65536), want.unsqueeze(0).expand(m // n, -1))


@pytest.mark.gpu
def test_concurrent_routed_batches_of_different_sizes_on_one_stream(gpu):
    """Two host threads send CHIP_FMT_DETECT batches of different sizes to the same stream at the same time.  The router's index
    lists and counters belong to (device, stream): taking them, the counters' reset, the router and both decoders' launches are one
    critical section, so neither batch may decode the other's lists (the race survived on this path).  Results are
    compared with the payload and with a single-threaded run."""
    import threading

    import compu_amd
    from bench_support import synth

    dev = "cuda:0"
    sizes = [320, 1100]
    jobs = []
    for k, n in enumerate(sizes):
        pay = synth.payloads(n, first_unit=1000 * k)
        packed, offs, lens = synth.mixed_units(pay, n, first_unit=1000 * k)
        jobs.append(dict(
            n=n, want=gpu.from_numpy(pay).to(dev), d_in=gpu.from_numpy(packed).to(dev), d_off=gpu.from_numpy(offs.astype(np.int64)).to(dev),
            d_len=gpu.from_numpy(lens.astype(np.int32)).to(dev), ooff=gpu.arange(n, dtype=gpu.int64, device=dev) * 65536,
            caps=gpu.full((n,), 65536, dtype=gpu.int32, device=dev), out=gpu.zeros(n * 65536, dtype=gpu.uint8, device=dev), lens=lens))
    single = []
    for j in jobs:  # the single-threaded answer
        ol, iu, st = compu_amd.decode_batch(0, j["d_in"], j["d_off"], j["d_len"], j["out"], j["ooff"], j["caps"])
        gpu.cuda.synchronize()
        assert bool((st == 2).all()) and gpu.equal(j["out"], j["want"])
        single.append((ol.clone(), iu.clone(), st.clone()))
        j["out"].zero_()
    res = [None, None]

    def work(k):
        j = jobs[k]
        for _ in range(25):
            res[k] = compu_amd.decode_batch(0, j["d_in"], j["d_off"], j["d_len"], j["out"], j["ooff"], j["caps"])

    compu_amd.trim()  # the lists are allocated (and grown by the larger batch) while both threads launch
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    gpu.cuda.synchronize()
    for k, j in enumerate(jobs):
        ol, iu, st = res[k]
        assert gpu.equal(st, single[k][2]) and gpu.equal(ol, single[k][0]) and gpu.equal(iu, single[k][1]), k
        assert gpu.equal(j["out"], j["want"]), k


def test_the_two_kernel_pipeline_passes_the_same_parity_tests(gpu):
    """CHIP_INFLATE_PIPE=1 (read once per process) sends raw-DEFLATE / zlib / gzip batches through tokens_kernel + lz77_kernel with
    the one-kernel path as the arbiter of everything unusual.  It is not the default -- it measured
    slower -- but it stays correct: the parity tests of this file that exercise whole batches, damaged and truncated streams, deep
    codes, small output ranges and multi-block streams run once more in a child process with the switch on."""
    import os
    import subprocess
    import sys

    if os.environ.get("CHIP_INFLATE_PIPE") == "1":
        pytest.skip("already inside the pipeline run")
    env = dict(os.environ, CHIP_INFLATE_PIPE="1")
    pick = "small_units or 64k_synthetic or deep_huffman or truncated_corrupt or large_multiblock or compu_status_flag"
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_inflate_gpu.py"), "-x", "-q", "-m", "gpu", "-k", pick, "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
