"""compu_amd -- MI355X (gfx950) batched compression backend behind compu's Decoder/Encoder surface.

The product is the C-ABI library ``libcompu_hip.so`` (see include/compu_hip.h).  This package is
the thin host-side mirror used by the tests and bench: the same names, argument meaning and status
contract as compu's ``decoder::Interface`` / ``Decoder`` / ``DecodeStatus`` (src/decoder/mod.rs)
and ``encoder::Interface`` / ``Encoder`` / ``EncodeOp`` / ``EncodeStatus`` (src/encoder/mod.rs),
bound with ctypes.  There is no CPU codec in here: without the HIP library (or without a GPU)
construction fails loudly.
"""
from .api import (  # noqa: F401
    Buffer,
    DeviceBuffer,
    PinnedBuffer,
    Decode,
    DecodeError,
    DecodeStatus,
    Decoder,
    Detection,
    Encode,
    EncodeOp,
    F_COMPU_STATUS,
    EncodeStatus,
    Encoder,
    Vec,
    ZlibMode,
    ZlibOptions,
    ZlibStrategy,
    ZstdOptions,
    decode_batch,
    decode_batch_host,
    decode_batch_multi,
    encode_batch_host,
    trim,
    decoder_interface,
    detect_batch,
    encode_batch,
    encode_bound,
    encoder_interface,
    lib,
    lib_path,
)
