//! `hip` interface implementation: zlib / gzip / raw deflate and zstd decoding on an MI355X through `libcompu_hip.so`.
//!
//! Same shape as `zlib_ng.rs`: one static vtable per format family, the state pointer is the backend's opaque
//! decoder object, created by the constructor and freed exactly once by `drop_fn`.

use core::ptr;

use super::zlib_common::ZlibMode;
use super::zstd::ZstdOptions;
use super::{Decode, DecodeError, DecodeStatus, Decoder, Interface};
use crate::hip_sys as sys;

static HIP_ZLIB: Interface = Interface {
    drop_fn,
    reset_fn,
    decode_fn,
    describe_error_fn: describe_zlib_error_fn,
};

static HIP_ZSTD: Interface = Interface {
    drop_fn,
    reset_fn,
    decode_fn,
    describe_error_fn: describe_zstd_error_fn,
};

impl Interface {
    ///Creates decoder with `hip` interface for zlib family of formats (same modes as `zlib_ng`).
    ///
    ///Returns `None` if unable to initialize it (no usable GPU, or lack of memory)
    pub fn zlib_hip(mode: ZlibMode) -> Option<Decoder> {
        crate::mem::hip_install_allocator();
        let opts = sys::chip_decoder_opts {
            window_log_max: 0,
            device: -1,
        };
        //`ZlibMode::max_bits()` is zlib's windowBits value (-15 / 15 / 31 / 47), which is the backend's format tag
        let instance = unsafe { sys::chip_decoder_new(mode.max_bits() as _, &opts) };
        ptr::NonNull::new(instance as *mut u8).map(|instance| HIP_ZLIB.inner_decoder(instance))
    }

    ///Creates decoder with `hip` interface for zstd.
    ///
    ///Returns `None` if unable to initialize it (no usable GPU, or lack of memory)
    pub fn zstd_hip(opts: ZstdOptions) -> Option<Decoder> {
        crate::mem::hip_install_allocator();
        let opts = sys::chip_decoder_opts {
            //`window_log` is private to `zstd.rs`; the maintainer adds `pub(super) const fn window_log_max(&self) -> i32`
            //next to `ZstdOptions::apply` (src/decoder/zstd.rs:50-74)
            window_log_max: opts.window_log_max(),
            device: -1,
        };
        let instance = unsafe { sys::chip_decoder_new(sys::CHIP_FMT_ZSTD, &opts) };
        ptr::NonNull::new(instance as *mut u8).map(|instance| HIP_ZSTD.inner_decoder(instance))
    }
}

#[inline]
unsafe fn decode_fn(state: ptr::NonNull<u8>, input: *const u8, input_remain: usize, output: *mut u8, output_remain: usize) -> Decode {
    let result = sys::chip_decode(state.as_ptr() as *mut sys::chip_decoder, input, input_remain, output, output_remain);
    Decode {
        input_remain: result.input_remain,
        output_remain: result.output_remain,
        status: match result.err {
            0 => Ok(match result.status {
                0 => DecodeStatus::NeedInput,
                1 => DecodeStatus::NeedOutput,
                _ => DecodeStatus::Finished,
            }),
            //same codes as the CPU backends: zlib's negative return values, -(ZSTD_ErrorCode)
            code => Err(DecodeError(code)),
        },
    }
}

#[inline]
fn reset_fn(state: ptr::NonNull<u8>) -> Option<ptr::NonNull<u8>> {
    let result = unsafe { sys::chip_decoder_reset(state.as_ptr() as *mut sys::chip_decoder) };
    ptr::NonNull::new(result as *mut u8)
}

#[inline]
fn drop_fn(state: ptr::NonNull<u8>) {
    unsafe {
        sys::chip_decoder_free(state.as_ptr() as *mut sys::chip_decoder);
    }
}

#[inline]
fn describe_zlib_error_fn(code: i32) -> Option<&'static str> {
    let result = unsafe { sys::chip_decoder_strerror(ZlibMode::Auto.max_bits() as _, code) };
    crate::utils::convert_c_str(result)
}

#[inline]
fn describe_zstd_error_fn(code: i32) -> Option<&'static str> {
    let result = unsafe { sys::chip_decoder_strerror(sys::CHIP_FMT_ZSTD, code) };
    crate::utils::convert_c_str(result)
}
