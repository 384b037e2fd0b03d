"""ctypes access to the system libzstd (build-container cross-check and input generation only)."""
import ctypes as C
import ctypes.util


class _Buf(C.Structure):
    _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


def load():
    for name in ("libzstd.so.1", ctypes.util.find_library("zstd")):
        if not name:
            continue
        try:
            z = C.CDLL(name)
        except OSError:
            continue
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        z.ZSTD_compress.restype = C.c_size_t
        z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        z.ZSTD_isError.argtypes = [C.c_size_t]
        z.ZSTD_getErrorCode.argtypes = [C.c_size_t]
        z.ZSTD_createDStream.restype = C.c_void_p
        z.ZSTD_freeDStream.argtypes = [C.c_void_p]
        z.ZSTD_decompressStream.restype = C.c_size_t
        z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.POINTER(_Buf), C.POINTER(_Buf)]
        z.ZSTD_createCCtx.restype = C.c_void_p
        z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        z.ZSTD_compress2.restype = C.c_size_t
        z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        return z
    return None


def compress(z, data, level=3, checksum=True, content_size=True):
    """One frame; checksum/content-size flags via the advanced API."""
    data = bytes(data)
    cctx = z.ZSTD_createCCtx()
    z.ZSTD_CCtx_setParameter(cctx, 100, level)  # ZSTD_c_compressionLevel
    z.ZSTD_CCtx_setParameter(cctx, 201, 1 if checksum else 0)  # ZSTD_c_checksumFlag
    z.ZSTD_CCtx_setParameter(cctx, 200, 1 if content_size else 0)  # ZSTD_c_contentSizeFlag
    cap = z.ZSTD_compressBound(len(data))
    dst = C.create_string_buffer(cap)
    src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
    n = z.ZSTD_compress2(cctx, dst, cap, src, len(data))
    z.ZSTD_freeCCtx(cctx)
    assert not z.ZSTD_isError(n)
    return dst.raw[:n]


def stream_decode_once(z, comp, out_cap):
    """One ZSTD_decompressStream call mapped the way src/decoder/zstd.rs:113-135 maps it.
    -> (output bytes, input_remain, output_remain, status or None, err)"""
    ds = z.ZSTD_createDStream()
    src = C.create_string_buffer(bytes(comp), len(comp)) if comp else C.create_string_buffer(1)
    dst = C.create_string_buffer(max(out_cap, 1))
    ib = _Buf(C.cast(src, C.c_void_p), len(comp), 0)
    ob = _Buf(C.cast(dst, C.c_void_p), out_cap, 0)
    ret = z.ZSTD_decompressStream(ds, C.byref(ob), C.byref(ib))
    out = dst.raw[: ob.pos]
    if ret == 0:
        st, err = 2, 0
    elif ob.pos == ob.size:
        st, err = 1, 0
    elif not z.ZSTD_isError(ret):
        st, err = 0, 0
    else:
        st, err = None, -z.ZSTD_getErrorCode(ret)
    z.ZSTD_freeDStream(ds)
    return out, len(comp) - ib.pos, out_cap - ob.pos, st, err


def craft_offset_frame(raw_blocks, block_len, offset, window_desc=0x00, fill=None):
    """A frame written by hand (RFC 8878): `raw_blocks` Raw_Blocks of `block_len` bytes, then one Compressed_Block with no literals
    and one sequence (literal length 0, match length 3, the given offset > 3 bytes back; all three tables in RLE mode, so the
    bitstream holds only the offset's extra bits).  No checksum, no content size; window from `window_desc` (0 = 1 KiB).
    -> (frame, the data of the raw blocks)"""
    import random

    rnd = random.Random(raw_blocks * 131 + block_len)
    data = fill if fill is not None else bytes(rnd.randrange(256) for _ in range(raw_blocks * block_len))
    out = bytearray(b"\x28\xb5\x2f\xfd") + bytes([0x00, window_desc])
    for k in range(raw_blocks):
        out += (block_len << 3).to_bytes(3, "little") + data[k * block_len:(k + 1) * block_len]
    value = offset + 3                      # Offset_Value; > 3: a new offset
    code = value.bit_length() - 1
    bits = (value - (1 << code)) | (1 << code)   # the extra bits, then the closing 1 bit above them
    stream = bits.to_bytes((code + 1 + 7) // 8, "little")
    body = bytes([0x00, 0x01, 0x54, 0x00, code, 0x00]) + stream  # raw literals of size 0; one sequence; RLE x 3: LL code 0, OF code, ML code 0
    out += ((len(body) << 3) | (2 << 1) | 1).to_bytes(3, "little") + body
    return bytes(out), data
