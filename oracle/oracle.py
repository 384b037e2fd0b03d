"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package compu_amd never does (see oracle/oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

NEED_INPUT, NEED_OUTPUT, FINISHED = 0, 1, 2  # DecodeStatus, src/decoder/mod.rs:139-146
MODE_DEFLATE, MODE_ZLIB, MODE_GZIP, MODE_AUTO = -15, 15, 31, 47  # src/decoder/zlib_common.rs:4-15
OP_PROCESS, OP_FLUSH, OP_FINISH = 0, 1, 2  # EncodeOp, src/encoder/mod.rs:12-23
ENC_CONTINUE, ENC_NEED_OUTPUT, ENC_FINISHED, ENC_ERROR = 0, 1, 2, 3  # src/encoder/mod.rs:27-38


class DecodeT(C.Structure):
    _fields_ = [("input_remain", C.c_size_t), ("output_remain", C.c_size_t), ("status", C.c_int32), ("err", C.c_int32)]


class EncodeT(C.Structure):
    _fields_ = [("input_remain", C.c_size_t), ("output_remain", C.c_size_t), ("status", C.c_int32)]


def build(force=False):
    """Compile oracle/*.c into liboracle.so (gcc only, no GPU involved)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c") or f.endswith(".h")]
    if not force and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        # ORACLE_LIB selects another build of the same sources (`make -C oracle asan` + LD_PRELOAD of libasan: the sanitizer run)
        path = os.environ.get("ORACLE_LIB")
        if not path:
            build()
            path = _LIB_PATH
        L = C.CDLL(path)
        u8p = C.c_void_p
        L.orc_inflate_new.restype = C.c_void_p
        L.orc_inflate_new.argtypes = [C.c_int]
        L.orc_inflate_decode.restype = DecodeT
        L.orc_inflate_decode.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, C.c_size_t]
        L.orc_inflate_reset.argtypes = [C.c_void_p]
        L.orc_inflate_free.argtypes = [C.c_void_p]
        L.orc_inflate_msg.restype = C.c_char_p
        L.orc_inflate_msg.argtypes = [C.c_void_p]
        L.orc_zlib_strerror.restype = C.c_char_p
        L.orc_zlib_strerror.argtypes = [C.c_int32]
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_uint32, u8p, C.c_size_t]
        L.orc_adler32.restype = C.c_uint32
        L.orc_adler32.argtypes = [C.c_uint32, u8p, C.c_size_t]
        units_args = [C.c_size_t, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p, C.c_int]
        L.orc_inflate_units.restype = C.c_size_t
        L.orc_inflate_units.argtypes = [C.c_int] + units_args
        if hasattr(L, "orc_zstd_new"):
            L.orc_zstd_new.restype = C.c_void_p
            L.orc_zstd_new.argtypes = [C.c_int]
            L.orc_zstd_decode.restype = DecodeT
            L.orc_zstd_decode.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, C.c_size_t]
            L.orc_zstd_reset.argtypes = [C.c_void_p]
            L.orc_zstd_free.argtypes = [C.c_void_p]
            L.orc_zstd_strerror.restype = C.c_char_p
            L.orc_zstd_strerror.argtypes = [C.c_int32]
            L.orc_xxh64.restype = C.c_uint64
            L.orc_xxh64.argtypes = [u8p, C.c_size_t, C.c_uint64]
            L.orc_zstd_units.restype = C.c_size_t
            L.orc_zstd_units.argtypes = units_args
        if hasattr(L, "orc_deflate_new"):
            L.orc_deflate_new.restype = C.c_void_p
            L.orc_deflate_new.argtypes = [C.c_int, C.c_int]
            L.orc_deflate_encode.restype = EncodeT
            L.orc_deflate_encode.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, C.c_size_t, C.c_int]
            L.orc_deflate_set_strategy.argtypes = [C.c_void_p, C.c_int]
            L.orc_deflate_reset.argtypes = [C.c_void_p]
            L.orc_deflate_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _StreamDecoder:
    """Shape of compu's Decoder (src/decoder/mod.rs:269-455) over an oracle state."""

    _new = _decode = _reset = _free = None

    def __init__(self, handle):
        if not handle:
            raise MemoryError("oracle decoder construction failed")
        self._h = handle

    def decode(self, data, out_len):
        """-> (produced bytes, input_remain, output_remain, status, err); status None when err != 0."""
        src = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
        dst = np.zeros(max(out_len, 1), np.uint8)
        r = self._decode(self._h, _ptr(src), len(data), _ptr(dst), out_len)
        produced = bytes(dst[: out_len - r.output_remain])
        return produced, r.input_remain, r.output_remain, (None if r.err else r.status), r.err

    def reset(self):
        self._reset(self._h)

    def close(self):
        if self._h:
            self._free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class InflateDecoder(_StreamDecoder):
    def __init__(self, mode=MODE_AUTO):
        L = lib()
        self._decode, self._reset, self._free = L.orc_inflate_decode, L.orc_inflate_reset, L.orc_inflate_free
        super().__init__(L.orc_inflate_new(mode))

    def msg(self):
        m = lib().orc_inflate_msg(self._h)
        return m.decode() if m else None


class ZstdDecoder(_StreamDecoder):
    def __init__(self, window_log_max=0):
        L = lib()
        self._decode, self._reset, self._free = L.orc_zstd_decode, L.orc_zstd_reset, L.orc_zstd_free
        super().__init__(L.orc_zstd_new(window_log_max))


class DeflateEncoder:
    """Shape of compu's Encoder (src/encoder/mod.rs:148-323) over the oracle deflater."""

    def __init__(self, mode=MODE_GZIP, level=1, strategy=0):
        self._h = lib().orc_deflate_new(mode, level)
        if not self._h:
            raise MemoryError("oracle encoder construction failed")
        if lib().orc_deflate_set_strategy(self._h, strategy) != 0:
            raise ValueError("bad strategy")

    def encode(self, data, out_len, op):
        src = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
        dst = np.zeros(max(out_len, 1), np.uint8)
        r = lib().orc_deflate_encode(self._h, _ptr(src), len(data), _ptr(dst), out_len, op)
        return bytes(dst[: out_len - r.output_remain]), r.input_remain, r.output_remain, r.status

    def reset(self):
        lib().orc_deflate_reset(self._h)

    def close(self):
        if self._h:
            lib().orc_deflate_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_all(dec, data, out_cap):
    """Drive a decoder to the end the way decode_vec_full does (src/decoder/mod.rs:360-385)."""
    out = b""
    pos = 0
    while True:
        got, in_rem, _out_rem, st, err = dec.decode(data[pos:], out_cap)
        out += got
        pos = len(data) - in_rem
        if err or st != NEED_OUTPUT:
            return out, in_rem, st, err
        if not got and in_rem == len(data) - pos and out_cap == 0:
            return out, in_rem, st, err


def crc32(data, crc=0):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return lib().orc_crc32(crc, _ptr(a), len(data))


def adler32(data, adler=1):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return lib().orc_adler32(adler, _ptr(a), len(data))


def xxh64(data, seed=0):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return lib().orc_xxh64(_ptr(a), len(data), seed)


def _units(fn, mode, in_buf, in_off, in_len, out_cap_total, out_off, out_cap, threads, out=None):
    n = len(in_len)
    in_buf = np.ascontiguousarray(in_buf, dtype=np.uint8)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    if out is None:
        out = np.zeros(max(int(out_cap_total), 1), np.uint8)
    out_len = np.zeros(n, np.uint32)
    status = np.zeros(n, np.int32)
    args = [n, _ptr(in_buf), _ptr(in_off), _ptr(in_len), _ptr(out), _ptr(out_off), _ptr(out_cap), _ptr(out_len), _ptr(status), threads]
    bad = fn(*([mode] + args if mode is not None else args))
    return out, out_len, status, bad


def inflate_units(mode, in_buf, in_off, in_len, out_cap_total, out_off, out_cap, threads=1, out=None):
    """Batch form of the compu CPU loop (one decoder per worker, reset per unit)."""
    return _units(lib().orc_inflate_units, mode, in_buf, in_off, in_len, out_cap_total, out_off, out_cap, threads, out)


def zstd_units(in_buf, in_off, in_len, out_cap_total, out_off, out_cap, threads=1, out=None):
    return _units(lib().orc_zstd_units, None, in_buf, in_off, in_len, out_cap_total, out_off, out_cap, threads, out)
