"""Host-side mirror of compu's Decoder / Encoder surface over libcompu_hip.so (ctypes).

Names follow the reference: decoder::{Interface, Decoder, Decode, DecodeStatus, DecodeError,
Detection, ZlibMode, ZstdOptions} (src/decoder/mod.rs, zlib_common.rs, zstd.rs),
encoder::{Interface, Encoder, Encode, EncodeOp, EncodeStatus, ZlibOptions} (src/encoder/mod.rs,
zlib_common.rs) and Buffer (src/buffer.rs).  All codec work happens in the HIP library.
"""
import ctypes as C
import enum
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libcompu_hip.so"


def lib_path():
    # COMPU_HIP_LIB selects another build of the same library (diagnostic builds); never a CPU stand-in
    return os.environ.get("COMPU_HIP_LIB") or os.path.join(_HERE, _LIB_NAME)


class _DecodeResult(C.Structure):
    _fields_ = [("input_remain", C.c_size_t), ("output_remain", C.c_size_t), ("status", C.c_int32), ("err", C.c_int32)]


class _EncodeResult(C.Structure):
    _fields_ = [("input_remain", C.c_size_t), ("output_remain", C.c_size_t), ("status", C.c_int32)]


class _DecoderOpts(C.Structure):
    _fields_ = [("window_log_max", C.c_int32), ("device", C.c_int32)]


class _EncoderOpts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("compression", C.c_int32), ("device", C.c_int32), ("strategy", C.c_int32), ("mem_level", C.c_int32)]


_lib = None


def lib():
    """Load libcompu_hip.so.  Raises if it has not been built: there is no fallback codec."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with compu_amd/csrc/build.sh (or __graft_entry__.build()); "
            "compu_amd has no CPU codec to fall back to"
        )
    try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    L = C.CDLL(path)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
    L.chip_device_count.restype = C.c_int
    L.chip_set_device.argtypes = [C.c_int]
    L.chip_version.restype = C.c_char_p
    L.chip_device_alloc.restype = vp
    L.chip_device_alloc.argtypes = [sz]
    L.chip_device_free.argtypes = [vp]
    L.chip_pinned_alloc.restype = vp
    L.chip_pinned_alloc.argtypes = [sz]
    L.chip_pinned_free.argtypes = [vp]
    L.chip_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    L.chip_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    L.chip_stream_sync.argtypes = [vp]
    L.chip_decode_batch_host.restype = C.c_int
    L.chip_decode_batch_host.argtypes = [C.c_int, C.c_size_t] + [vp] * 9 + [C.c_int, C.c_size_t]
    L.chip_decode_batch_multi.restype = C.c_int
    L.chip_decode_batch_multi.argtypes = [C.c_int, C.c_size_t] + [vp] * 9 + [vp, C.c_int, C.c_size_t]
    L.chip_partition_units.restype = C.c_int
    L.chip_partition_units.argtypes = [C.c_size_t, vp, vp, C.c_int, vp]
    L.chip_encode_batch_host.restype = C.c_int
    L.chip_encode_batch_host.argtypes = [C.c_int, C.c_int, C.c_size_t] + [vp] * 8 + [C.c_int, C.c_size_t]
    L.chip_trim.restype = C.c_int
    L.chip_trim.argtypes = []
    L.chip_decoder_new.restype = vp
    L.chip_decoder_new.argtypes = [C.c_int, C.POINTER(_DecoderOpts)]
    L.chip_decode.restype = _DecodeResult
    L.chip_decode.argtypes = [vp, vp, sz, vp, sz]
    L.chip_decoder_reset.restype = vp
    L.chip_decoder_reset.argtypes = [vp]
    L.chip_decoder_free.argtypes = [vp]
    L.chip_decoder_footprint.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.chip_decoder_footprint.restype = None
    L.chip_set_allocator.argtypes = [vp, vp, vp]
    L.chip_set_allocator.restype = None
    L.chip_decoder_strerror.restype = C.c_char_p
    L.chip_decoder_strerror.argtypes = [C.c_int, i32]
    L.chip_decode_batch.restype = C.c_int
    L.chip_decode_batch.argtypes = [C.c_int, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.chip_decode_batch_ex.restype = C.c_int
    L.chip_decode_batch_ex.argtypes = [C.c_int, C.c_uint32, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.chip_detect.restype = C.c_int
    L.chip_detect.argtypes = [vp, sz]
    L.chip_detect_batch.restype = C.c_int
    L.chip_detect_batch.argtypes = [sz, vp, vp, vp, vp, vp]
    L.chip_encoder_new.restype = vp
    L.chip_encoder_new.argtypes = [C.POINTER(_EncoderOpts)]
    L.chip_encode.restype = _EncodeResult
    L.chip_encode.argtypes = [vp, vp, sz, vp, sz, C.c_int]
    L.chip_encoder_reset.restype = vp
    L.chip_encoder_reset.argtypes = [vp]
    L.chip_encoder_free.argtypes = [vp]
    L.chip_encode_batch.restype = C.c_int
    L.chip_encode_batch.argtypes = [C.c_int, C.c_int, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.chip_encode_batch_ex.restype = C.c_int
    L.chip_encode_batch_ex.argtypes = [C.c_int, C.c_int, C.c_int, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.chip_encode_bound.restype = sz
    L.chip_encode_bound.argtypes = [C.c_int, sz]
    _lib = L
    return L


# ---- enums and result types ----------------------------------------------------------------


class DecodeStatus(enum.IntEnum):
    """src/decoder/mod.rs:139-146"""

    NeedInput = 0
    NeedOutput = 1
    Finished = 2


class DecodeError(Exception):
    """src/decoder/mod.rs:117-135: transparent wrapper of the backend's i32 code."""

    def __init__(self, code=0):
        super().__init__(code)
        self.code = int(code)

    @classmethod
    def no_error(cls):
        return cls(0)

    def as_raw(self):
        return self.code

    def __eq__(self, other):
        return isinstance(other, DecodeError) and other.code == self.code

    def __hash__(self):
        return hash(("DecodeError", self.code))

    def __repr__(self):
        return f"DecodeError({self.code})"


class Decode:
    """src/decoder/mod.rs:150-157; ``status`` is a DecodeStatus (Ok) or a DecodeError (Err)."""

    __slots__ = ("input_remain", "output_remain", "status")

    def __init__(self, input_remain, output_remain, status):
        self.input_remain = input_remain
        self.output_remain = output_remain
        self.status = status

    def is_ok(self):
        return isinstance(self.status, DecodeStatus)

    def __repr__(self):
        return f"Decode(input_remain={self.input_remain}, output_remain={self.output_remain}, status={self.status!r})"


class ZlibMode(enum.IntEnum):
    """src/decoder/zlib_common.rs:4-15 (decoder default Auto) / src/encoder/zlib_common.rs:28-37"""

    Deflate = -15
    Zlib = 15
    Gzip = 31
    Auto = 47


FMT_ZSTD = 100


class ZstdOptions:
    """src/decoder/zstd.rs:22-74"""

    def __init__(self):
        self._window_log = 0

    def window_log(self, window_log):
        assert 10 <= window_log <= 31  # ZSTD_WINDOWLOG_MIN .. ZSTD_WINDOWLOG_MAX_64, zstd.rs:40-47
        self._window_log = window_log
        return self


class EncodeOp(enum.IntEnum):
    """src/encoder/mod.rs:12-23"""

    Process = 0
    Flush = 1
    Finish = 2


class EncodeStatus(enum.IntEnum):
    """src/encoder/mod.rs:27-38"""

    Continue = 0
    NeedOutput = 1
    Finished = 2
    Error = 3


class Encode:
    """src/encoder/mod.rs:42-49"""

    __slots__ = ("input_remain", "output_remain", "status")

    def __init__(self, input_remain, output_remain, status):
        self.input_remain = input_remain
        self.output_remain = output_remain
        self.status = status

    def __repr__(self):
        return f"Encode(input_remain={self.input_remain}, output_remain={self.output_remain}, status={self.status!r})"


class ZlibStrategy(enum.IntEnum):
    """src/encoder/zlib_common.rs:5-24"""

    Default = 0
    Filtered = 1
    HuffmanOnly = 2
    Rle = 3
    Fixed = 4


class ZlibOptions:
    """src/encoder/zlib_common.rs:47-103 (defaults: Gzip, Default strategy, mem_level 8, level 9: zlib_common.rs:59-66)"""

    def __init__(self):
        self._mode = ZlibMode.Gzip
        self._compression = 9
        self._strategy = ZlibStrategy.Default
        self._mem_level = 8

    def strategy(self, strategy):
        self._strategy = ZlibStrategy(strategy)
        return self

    def mem_level(self, mem_level):
        # the reference's setter asserts `mem_level > MAX_MEM_LEVEL` (an inverted check, zlib_common.rs:88); the value that
        # reaches deflateInit2_ must be 1..9, which is what the backend accepts
        assert 0 < mem_level <= 9
        self._mem_level = mem_level
        return self

    def mode(self, mode):
        assert mode in (ZlibMode.Deflate, ZlibMode.Zlib, ZlibMode.Gzip)
        self._mode = ZlibMode(mode)
        return self

    def compression(self, level):
        assert -1 <= level <= 9  # -1 = zlib's default (zlib_common.rs:96-103)
        self._compression = level
        return self


class Detection(enum.IntEnum):
    """src/decoder/mod.rs:9-21; detect() returns None when there are too few bytes (mod.rs:97-104)."""

    Zstd = 0
    Gzip = 1
    Zlib = 2
    Unknown = 3

    @staticmethod
    def detect(data):
        data = bytes(data)
        k = lib().chip_detect(data if data else b"\0", len(data))
        return None if k < 0 else Detection(k)


# ---- Vec<u8> stand-in so decode_vec / decode_vec_full read like the reference -----------------


class Vec:
    """A byte vector with explicit capacity (Rust's Vec<u8>: len() <= capacity())."""

    def __init__(self, capacity=0):
        import numpy as np

        self._buf = np.zeros(capacity, np.uint8)
        self._len = 0

    @classmethod
    def with_capacity(cls, capacity):
        return cls(capacity)

    def __len__(self):
        return self._len

    def capacity(self):
        return self._buf.size

    def try_reserve_exact(self, additional):
        import numpy as np

        need = self._len + additional
        if need > self._buf.size:
            grown = np.zeros(need, np.uint8)
            grown[: self._len] = self._buf[: self._len]
            self._buf = grown

    reserve = try_reserve_exact

    def spare_capacity_len(self):
        return self._buf.size - self._len

    def set_len(self, n):
        assert n <= self._buf.size
        self._len = n

    def clear(self):
        self._len = 0

    def truncate(self, n):
        self._len = min(self._len, n)

    def extend_from_slice(self, data):
        import numpy as np

        self.try_reserve_exact(len(data))
        self._buf[self._len : self._len + len(data)] = np.frombuffer(bytes(data), dtype=np.uint8)
        self._len += len(data)

    def __bytes__(self):
        return self._buf[: self._len].tobytes()

    def __eq__(self, other):
        return bytes(self) == bytes(other)


def _in_ptr(data):
    """(keepalive, pointer, length) for a read-only bytes-like; never a NULL pointer (mod.rs:283)."""
    import numpy as np

    if not isinstance(data, (bytes, bytearray)):
        data = bytes(memoryview(data).cast("B"))
    n = len(data)
    if n == 0:
        keep = np.zeros(1, np.uint8)
        return keep, C.c_void_p(keep.ctypes.data), 0
    keep = np.frombuffer(data, dtype=np.uint8)
    return keep, C.c_void_p(keep.ctypes.data), n


def _out_ptr(buf, offset, length):
    import numpy as np

    if length == 0:
        keep = np.zeros(1, np.uint8)
        return keep, C.c_void_p(keep.ctypes.data)
    keep = buf if isinstance(buf, np.ndarray) else np.frombuffer(buf, dtype=np.uint8)
    assert keep.flags.writeable and offset + length <= keep.size
    return keep, C.c_void_p(keep.ctypes.data + offset)


# ---- Decoder ---------------------------------------------------------------------------------


class Decoder:
    """src/decoder/mod.rs:269-455 over a chip_decoder instance."""

    def __init__(self, handle, fmt):
        self._h = handle
        self._fmt = fmt

    # raw_decode / decode, mod.rs:290-317
    def decode(self, input, output, out_offset=0, out_len=None):
        """Decode `input` into the writable buffer `output[out_offset : out_offset+out_len]`."""
        if out_len is None:
            out_len = len(output) - out_offset
        k1, ip, n = _in_ptr(input)
        k2, op = _out_ptr(output, out_offset, out_len)
        r = lib().chip_decode(self._h, ip, n, op, out_len)
        del k1, k2
        st = DecodeError(r.err) if r.err else DecodeStatus(r.status)
        return Decode(r.input_remain, r.output_remain, st)

    # mod.rs:323-335
    def decode_vec(self, input, output):
        spare = output.spare_capacity_len()
        result = self.decode(input, output._buf, len(output), spare)
        if result.is_ok():
            output.set_len(len(output) + spare - result.output_remain)
        return result

    # mod.rs:360-385
    def decode_vec_full(self, input, output):
        RESERVE_DEFAULT = 1024
        input = bytes(input)
        input_len = len(input)
        if input_len < RESERVE_DEFAULT:
            output.try_reserve_exact(input_len)
            reserve_size = input_len // 3
        elif input_len < RESERVE_DEFAULT * 16:
            output.try_reserve_exact(input_len + input_len // 3)
            reserve_size = RESERVE_DEFAULT
        else:
            output.try_reserve_exact(input_len * 2)
            reserve_size = RESERVE_DEFAULT * 8
        while True:
            result = self.decode_vec(input, output)
            if result.status == DecodeStatus.NeedOutput:
                input = input[len(input) - result.input_remain :]
                output.try_reserve_exact(reserve_size)
                continue
            return result

    # mod.rs:433-441
    def reset(self):
        h = lib().chip_decoder_reset(self._h)
        if h:
            self._h = h
            return True
        return False

    def footprint(self):
        """(pinned host bytes, device bytes) this decoder holds right now (chip_decoder_footprint)."""
        a, b = C.c_size_t(0), C.c_size_t(0)
        lib().chip_decoder_footprint(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    # mod.rs:445-447
    def describe_error(self, error):
        s = lib().chip_decoder_strerror(self._fmt, error.as_raw())
        return None if s is None else s.decode()

    def close(self):
        if self._h:
            lib().chip_decoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class decoder_interface:
    """decoder::Interface constructors of the `hip` variant."""

    @staticmethod
    def zlib_hip(mode=ZlibMode.Auto, device=-1):
        """Interface::zlib_ng(mode), src/decoder/zlib_ng.rs:61-90; None on failure."""
        opts = _DecoderOpts(0, device)
        h = lib().chip_decoder_new(int(mode), C.byref(opts))
        return Decoder(h, int(mode)) if h else None

    @staticmethod
    def zstd_hip(opts=None, device=-1):
        """Interface::zstd(opts), src/decoder/zstd.rs:81-94; None on failure."""
        o = _DecoderOpts(opts._window_log if opts else 0, device)
        h = lib().chip_decoder_new(FMT_ZSTD, C.byref(o))
        return Decoder(h, FMT_ZSTD) if h else None


# ---- Encoder ---------------------------------------------------------------------------------


class Encoder:
    """src/encoder/mod.rs:148-323 over a chip_encoder instance."""

    def __init__(self, handle):
        self._h = handle

    # raw_encode / encode, mod.rs:171-199
    def encode(self, input, output, op, out_offset=0, out_len=None):
        if out_len is None:
            out_len = len(output) - out_offset
        k1, ip, n = _in_ptr(input)
        k2, outp = _out_ptr(output, out_offset, out_len)
        r = lib().chip_encode(self._h, ip, n, outp, out_len, int(op))
        del k1, k2
        return Encode(r.input_remain, r.output_remain, EncodeStatus(r.status))

    # mod.rs:203-213 (sets the length even on Error)
    def encode_vec(self, input, output, op):
        spare = output.spare_capacity_len()
        result = self.encode(input, output._buf, op, len(output), spare)
        output.set_len(len(output) + spare - result.output_remain)
        return result

    # mod.rs:239-267: reserve policy, then loop on NeedOutput (and on Continue while finishing)
    def encode_vec_full(self, input, output, op):
        RESERVE_DEFAULT = 1024
        input = bytes(input)
        input_len = len(input)
        if input_len < RESERVE_DEFAULT:
            output.try_reserve_exact(input_len)
            reserve_size = input_len // 3
        elif input_len < RESERVE_DEFAULT * 16:
            output.try_reserve_exact(input_len // 2)
            reserve_size = RESERVE_DEFAULT
        else:
            output.try_reserve_exact(input_len // 3)
            reserve_size = RESERVE_DEFAULT * 8
        while True:
            result = self.encode_vec(input, output, op)
            if result.status == EncodeStatus.NeedOutput:
                input = input[len(input) - result.input_remain :]
                # the reference reserves `reserve_size` (0 for inputs under 3 bytes, where it would
                # spin); keep at least one byte of progress
                output.try_reserve_exact(max(reserve_size, 1))
                continue
            if result.status == EncodeStatus.Continue and op == EncodeOp.Finish:
                input = input[len(input) - result.input_remain :]
                continue
            return result

    # mod.rs:314-321
    def reset(self):
        h = lib().chip_encoder_reset(self._h)
        if h:
            self._h = h
            return True
        return False

    def close(self):
        if self._h:
            lib().chip_encoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class encoder_interface:
    @staticmethod
    def zlib_hip(opts=None, device=-1):
        """Interface::zlib_ng(opts), src/encoder/zlib_ng.rs:50-87; None on failure."""
        opts = opts or ZlibOptions()
        o = _EncoderOpts(int(opts._mode), opts._compression, device, int(opts._strategy), opts._mem_level)
        h = lib().chip_encoder_new(C.byref(o))
        return Encoder(h) if h else None


# ---- Buffer<N>, src/buffer.rs ---------------------------------------------------------------


class Buffer:
    """Fixed-size buffer with a cursor (src/buffer.rs:1-49)."""

    def __init__(self, n):
        import numpy as np

        self._buf = np.zeros(n, np.uint8)
        self.cursor = 0

    def data(self):
        return self._buf[: self.cursor].tobytes()

    def consume(self):
        self.cursor = 0

    # decoder/mod.rs:507-531
    def decode(self, decoder, input):
        spare = len(self._buf) - self.cursor
        result = decoder.decode(input, self._buf, self.cursor, spare)
        if isinstance(result.status, DecodeError):
            raise result.status
        self.cursor = self.cursor + spare - result.output_remain
        return len(input) - result.input_remain, result.status

    # encoder/mod.rs:395-412
    def encode(self, encoder, input, op):
        spare = len(self._buf) - self.cursor
        result = encoder.encode(input, self._buf, op, self.cursor, spare)
        self.cursor = self.cursor + spare - result.output_remain
        return len(input) - result.input_remain, result.status


class PinnedBuffer(Buffer):
    """Buffer<N>'s cursor API (src/buffer.rs:1-49) over page-locked host memory (chip_pinned_alloc = hipHostMalloc): the
    north star's pinned-host buffer type.  Used like Buffer with the streaming Decoder / Encoder."""

    def __init__(self, n):
        import numpy as np

        self._ptr = lib().chip_pinned_alloc(n)
        if not self._ptr:
            raise MemoryError("chip_pinned_alloc failed")
        self._buf = np.ctypeslib.as_array((C.c_uint8 * n).from_address(self._ptr))
        self.cursor = 0

    def close(self):
        if getattr(self, "_ptr", None):
            self._buf = None
            lib().chip_pinned_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """The north star's device buffer type: GPU memory (chip_device_alloc = hipMalloc, where src/mem.rs routes device
    allocations) with Buffer<N>'s cursor.  upload() appends host bytes, decode_batch() appends decoded units behind the
    cursor without the data touching the host, download() reads back."""

    def __init__(self, n):
        self._cap = n
        self._ptr = lib().chip_device_alloc(n + 16)
        if not self._ptr:
            raise MemoryError("chip_device_alloc failed")
        self.cursor = 0

    def ptr(self, offset=0):
        return self._ptr + offset

    def __len__(self):
        return self.cursor

    def capacity(self):
        return self._cap

    def spare_capacity_len(self):
        return self._cap - self.cursor

    def consume(self):
        self.cursor = 0

    def upload(self, data):
        k, ip, n = _in_ptr(data)
        if n > self._cap - self.cursor:
            raise ValueError("does not fit")
        L = lib()
        if n and (L.chip_memcpy_h2d(self._ptr + self.cursor, ip, n, None) != 0 or L.chip_stream_sync(None) != 0):
            raise RuntimeError("upload failed")
        self.cursor += n
        return n

    def download(self, offset=0, n=None):
        import numpy as np

        n = self.cursor - offset if n is None else n
        assert offset + n <= self.cursor
        out = np.empty(n, np.uint8)
        L = lib()
        if n and (L.chip_memcpy_d2h(out.ctypes.data, self._ptr + offset, n, None) != 0 or L.chip_stream_sync(None) != 0):
            raise RuntimeError("download failed")
        return out

    def decode_batch(self, fmt, src, in_off, in_len, out_off, out_cap, span):
        """chip_decode_batch from DeviceBuffer `src` into this buffer's spare capacity (out_off relative to it); the
        per-unit arrays are host sequences, uploaded here.  Returns (out_len, in_used, status) as numpy arrays."""
        import numpy as np

        n = len(in_len)
        if span > self._cap - self.cursor:
            raise ValueError("does not fit")
        host = np.zeros(n * 10 + 16, np.uint32)
        host[: 2 * n].view(np.uint64)[:] = np.asarray(in_off, np.uint64)
        host[2 * n : 4 * n].view(np.uint64)[:] = np.asarray(out_off, np.uint64)
        host[4 * n : 5 * n] = np.asarray(in_len, np.uint32)
        host[5 * n : 6 * n] = np.asarray(out_cap, np.uint32)
        arr = DeviceBuffer(host.nbytes)
        arr.upload(host)
        a = arr.ptr()
        L = lib()
        rc = L.chip_decode_batch(int(fmt), n, src.ptr(), a, a + 16 * n, self._ptr + self.cursor, a + 8 * n, a + 20 * n, a + 24 * n, a + 28 * n,
                                 a + 32 * n, None)
        if rc != 0 or L.chip_stream_sync(None) != 0:
            raise RuntimeError(f"chip_decode_batch failed: {rc}")
        self.cursor += span
        res = arr.download(24 * n, 12 * n).view(np.uint32)
        arr.close()
        return res[:n].copy(), res[n : 2 * n].copy(), res[2 * n :].view(np.int32).copy()

    def close(self):
        if getattr(self, "_ptr", None):
            lib().chip_device_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- batched entry points on torch CUDA tensors -----------------------------------------------


def _dp(t):
    return C.c_void_p(t.data_ptr())


def _stream_ptr(stream):
    if stream is None:
        import torch

        stream = torch.cuda.current_stream()
    return C.c_void_p(getattr(stream, "cuda_stream", stream))


def _check_tensors(pairs):
    """Every tensor handed to a device batch call: on one GPU, contiguous, of the dtype the kernel reads it as (a CPU
    tensor or int32 offsets would otherwise be read as device u64 offsets: a GPU fault instead of a Python error)."""
    dev = pairs[0][0].device
    for t, dt in pairs:
        if not (t.is_cuda and t.is_contiguous() and t.dtype == dt and t.device == dev):
            raise TypeError(f"expected a contiguous {dt} tensor on {dev}, got {t.dtype} on {t.device} (contiguous={t.is_contiguous()})")
    return dev


F_COMPU_STATUS = 1  # CHIP_F_COMPU_STATUS


def decode_batch(fmt, in_buf, in_off, in_len, out_buf, out_off, out_cap, out_len=None, in_used=None, status=None, stream=None, flags=0):
    """chip_decode_batch[_ex] on device tensors: in_buf/out_buf uint8, *_off int64 (read as u64),
    in_len/out_cap/out_len/in_used int32 (read as u32), status int32.  Only enqueues."""
    import torch

    n = in_len.numel()
    dev = in_buf.device
    if out_len is None:
        out_len = torch.empty(n, dtype=torch.int32, device=dev)
    if in_used is None:
        in_used = torch.empty(n, dtype=torch.int32, device=dev)
    if status is None:
        status = torch.empty(n, dtype=torch.int32, device=dev)
    _check_tensors(((in_buf, torch.uint8), (out_buf, torch.uint8), (in_off, torch.int64), (out_off, torch.int64), (in_len, torch.int32),
                    (out_cap, torch.int32), (out_len, torch.int32), (in_used, torch.int32), (status, torch.int32)))
    with torch.cuda.device(dev):  # scratch and the stream come from the tensors' device, not whatever is current
        rc = lib().chip_decode_batch_ex(int(fmt), int(flags), n, _dp(in_buf), _dp(in_off), _dp(in_len), _dp(out_buf), _dp(out_off), _dp(out_cap),
                                        _dp(out_len), _dp(in_used), _dp(status), _stream_ptr(stream))
    if rc != 0:
        raise RuntimeError(f"chip_decode_batch failed: {rc}")
    return out_len, in_used, status


def decode_batch_host(fmt, in_buf, in_off, in_len, out_buf, out_off, out_cap, device=-1, slice_bytes=0):
    """chip_decode_batch_host over numpy arrays in host memory (pinned or not): returns (out_len, in_used, status)."""
    import numpy as np

    n = len(in_len)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    out_len = np.zeros(n, np.uint32)
    in_used = np.zeros(n, np.uint32)
    status = np.zeros(n, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib().chip_decode_batch_host(int(fmt), n, p(in_buf), p(in_off), p(in_len), p(out_buf), p(out_off), p(out_cap), p(out_len), p(in_used),
                                      p(status), int(device), int(slice_bytes))
    if rc != 0:
        raise RuntimeError(f"chip_decode_batch_host failed: {rc}")
    return out_len, in_used, status


def decode_batch_multi(fmt, in_buf, in_off, in_len, out_buf, out_off, out_cap, devices=None, slice_bytes=0):
    """chip_decode_batch_multi: host-memory units partitioned over `devices` (None = every visible GPU), one host
    thread and two streams per device, a CHIP_FMT_DETECT batch bucketed by format first.  Returns (out_len, in_used, status)."""
    import numpy as np

    n = len(in_len)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    out_len = np.zeros(n, np.uint32)
    in_used = np.zeros(n, np.uint32)
    status = np.zeros(n, np.int32)
    devs = np.ascontiguousarray(devices if devices is not None else [], dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib().chip_decode_batch_multi(int(fmt), n, p(in_buf), p(in_off), p(in_len), p(out_buf), p(out_off), p(out_cap), p(out_len), p(in_used),
                                       p(status), p(devs) if len(devs) else None, len(devs), int(slice_bytes))
    if rc != 0:
        raise RuntimeError(f"chip_decode_batch_multi failed: {rc}")
    return out_len, in_used, status


def encode_batch_host(fmt, level, in_buf, in_off, in_len, out_buf, out_off, out_cap, device=-1, slice_bytes=0):
    """chip_encode_batch_host over numpy arrays in host memory: returns (out_len, status)."""
    import numpy as np

    n = len(in_len)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    out_len = np.zeros(n, np.uint32)
    status = np.zeros(n, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib().chip_encode_batch_host(int(fmt), int(level), n, p(in_buf), p(in_off), p(in_len), p(out_buf), p(out_off), p(out_cap), p(out_len),
                                      p(status), int(device), int(slice_bytes))
    if rc != 0:
        raise RuntimeError(f"chip_encode_batch_host failed: {rc}")
    return out_len, status


def trim():
    """Give the inflate kernel's cached token scratch of the current device back (chip_trim)."""
    rc = lib().chip_trim()
    if rc != 0:
        raise RuntimeError(f"chip_trim failed: {rc}")


def detect_batch(in_buf, in_off, in_len, kind=None, stream=None):
    import torch

    n = in_len.numel()
    if kind is None:
        kind = torch.empty(n, dtype=torch.int32, device=in_buf.device)
    dev = _check_tensors(((in_buf, torch.uint8), (in_off, torch.int64), (in_len, torch.int32), (kind, torch.int32)))
    with torch.cuda.device(dev):
        rc = lib().chip_detect_batch(n, _dp(in_buf), _dp(in_off), _dp(in_len), _dp(kind), _stream_ptr(stream))
    if rc != 0:
        raise RuntimeError(f"chip_detect_batch failed: {rc}")
    return kind


def encode_bound(fmt, in_len):
    return lib().chip_encode_bound(int(fmt), int(in_len))


def encode_batch(fmt, level, in_buf, in_off, in_len, out_buf, out_off, out_cap, out_len=None, status=None, stream=None, strategy=0):
    """chip_encode_batch_ex over device tensors: level 0 stored, 1 fixed Huffman, 2..9 (-1 = 6) dynamic Huffman blocks."""
    import torch

    n = in_len.numel()
    dev = in_buf.device
    if out_len is None:
        out_len = torch.empty(n, dtype=torch.int32, device=dev)
    if status is None:
        status = torch.empty(n, dtype=torch.int32, device=dev)
    _check_tensors(((in_buf, torch.uint8), (out_buf, torch.uint8), (in_off, torch.int64), (out_off, torch.int64), (in_len, torch.int32),
                    (out_cap, torch.int32), (out_len, torch.int32), (status, torch.int32)))
    with torch.cuda.device(dev):
        rc = lib().chip_encode_batch_ex(int(fmt), int(level), int(strategy), n, _dp(in_buf), _dp(in_off), _dp(in_len), _dp(out_buf), _dp(out_off),
                                        _dp(out_cap), _dp(out_len), _dp(status), _stream_ptr(stream))
    if rc != 0:
        raise RuntimeError(f"chip_encode_batch failed: {rc}")
    return out_len, status
