"""Synthetic inputs for the batched-decode benchmark and the large parity cases.

Payloads come from bench_support/synth.c (seeded, deterministic per unit index); compression of
the payloads uses Python's zlib module (system zlib 1.2.11 in this image) in a thread pool.  This
only prepares INPUTS -- nothing here is on the measured or shipped path.
"""
import ctypes as C
import os
import subprocess
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsynth.so")

# calibrated once against zlib level 6 on 64 KiB units: compressed/uncompressed = 0.505 +- 0.003
P_MATCH = 0.20
ZIPF_S = 1.2
LEN_P = 0.2
UNIT = 65536


def build(force=False):
    src = os.path.join(_HERE, "synth.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _LIB, src, "-lm", "-lpthread"])
    return _LIB


_lib = None


def _l():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.synth_units.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_int]
    return _lib


def payloads(n_units, first_unit=0, unit_size=UNIT, threads=None, p_match=P_MATCH, zipf_s=ZIPF_S, len_p=LEN_P):
    """uint8 array of n_units*unit_size bytes: units first_unit .. first_unit+n_units-1."""
    threads = threads or min(32, os.cpu_count() or 1)
    out = np.empty(n_units * unit_size, np.uint8)
    _l().synth_units(out.ctypes.data, first_unit, n_units, unit_size, p_match, zipf_s, len_p, threads)
    return out


def _deflate_one(args):
    data, kind, wbits = args
    if kind == "stored":
        co = zlib.compressobj(0, zlib.DEFLATED, wbits)
    elif kind == "fixed":
        co = zlib.compressobj(6, zlib.DEFLATED, wbits, 8, zlib.Z_FIXED)
    elif kind == "dynamic":
        co = zlib.compressobj(6, zlib.DEFLATED, wbits)
    elif kind == "level1":
        co = zlib.compressobj(1, zlib.DEFLATED, wbits)
    else:
        raise ValueError(kind)
    return co.compress(data) + co.flush()


def deflate_units(payload, n_units, unit_size=UNIT, kind="dynamic", wbits=-15, threads=None):
    """Compress each unit independently.  Returns (packed uint8 array padded to 4 B, offsets u64, lengths u32)."""
    threads = threads or min(32, os.cpu_count() or 1)
    mv = memoryview(payload)
    jobs = [(mv[i * unit_size : (i + 1) * unit_size], kind, wbits) for i in range(n_units)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(_deflate_one, jobs, chunksize=64))
    lens = np.array([len(p) for p in parts], dtype=np.uint32)
    offs = np.zeros(n_units, dtype=np.uint64)
    if n_units > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    total = int(lens.sum())
    packed = np.zeros((total + 3) & ~3, np.uint8)
    packed[:total] = np.frombuffer(b"".join(parts), dtype=np.uint8)
    return packed, offs, lens


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


_zstd = None


def _zstd_lib():
    global _zstd
    if _zstd is None:
        z = C.CDLL("libzstd.so.1")
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        z.ZSTD_createCCtx.restype = C.c_void_p
        z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        z.ZSTD_compress2.restype = C.c_size_t
        z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.ZSTD_isError.argtypes = [C.c_size_t]
        _zstd = z
    return _zstd


def _mixed_one(args):
    data, unit_index = args
    if _splitmix64(unit_index) & 1:
        co = zlib.compressobj(6, zlib.DEFLATED, 31)  # gzip, no name
        return co.compress(data) + co.flush()
    z = _zstd_lib()
    arr = np.frombuffer(data, dtype=np.uint8)
    cctx = z.ZSTD_createCCtx()
    z.ZSTD_CCtx_setParameter(cctx, 100, 3)  # level 3
    z.ZSTD_CCtx_setParameter(cctx, 201, 1)  # content checksum
    cap = z.ZSTD_compressBound(arr.size)
    dst = np.empty(cap, np.uint8)
    n = z.ZSTD_compress2(cctx, dst.ctypes.data, cap, arr.ctypes.data, arr.size)
    z.ZSTD_freeCCtx(cctx)
    assert not z.ZSTD_isError(n)
    return dst[:n].tobytes()


def mixed_units(payload, n_units, first_unit=0, unit_size=UNIT, threads=None):
    """BASELINE.json configs[4]: unit i is a gzip member (zlib level 6) if splitmix64(i) is odd, else a
    zstd frame (level 3, single block, with checksum).  Returns (packed, offsets, lengths)."""
    threads = threads or min(32, os.cpu_count() or 1)
    mv = memoryview(payload)
    jobs = [(mv[i * unit_size : (i + 1) * unit_size], first_unit + i) for i in range(n_units)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(_mixed_one, jobs, chunksize=64))
    lens = np.array([len(p) for p in parts], dtype=np.uint32)
    offs = np.zeros(n_units, dtype=np.uint64)
    if n_units > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    total = int(lens.sum())
    packed = np.zeros((total + 3) & ~3, np.uint8)
    packed[:total] = np.frombuffer(b"".join(parts), dtype=np.uint8)
    return packed, offs, lens
