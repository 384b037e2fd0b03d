//! Raw bindings of `libcompu_hip.so` (include/compu_hip.h), the MI355X backend.
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct chip_decoder {
    _private: [u8; 0],
}
#[repr(C)]
pub struct chip_encoder {
    _private: [u8; 0],
}

///`chip_decode_result`: `Decode` with the error code split out (`err == 0` means `Ok(status)`)
#[repr(C)]
pub struct chip_decode_result {
    pub input_remain: usize,
    pub output_remain: usize,
    pub status: i32,
    pub err: i32,
}

#[repr(C)]
pub struct chip_encode_result {
    pub input_remain: usize,
    pub output_remain: usize,
    pub status: i32,
}

#[repr(C)]
pub struct chip_decoder_opts {
    pub window_log_max: i32,
    pub device: i32,
}

#[repr(C)]
pub struct chip_encoder_opts {
    pub mode: i32,
    pub compression: i32,
    pub device: i32,
    pub strategy: i32,
    pub mem_level: i32,
}

pub const CHIP_FMT_ZSTD: c_int = 100;

pub type chip_malloc_fn = unsafe extern "C" fn(opaque: *mut c_void, size: usize) -> *mut c_void;
pub type chip_free_fn = unsafe extern "C" fn(opaque: *mut c_void, ptr: *mut c_void);

#[link(name = "compu_hip")]
extern "C" {
    pub fn chip_device_count() -> c_int;
    pub fn chip_set_allocator(malloc_fn: Option<chip_malloc_fn>, free_fn: Option<chip_free_fn>, opaque: *mut c_void);
    pub fn chip_device_alloc(size: usize) -> *mut c_void;
    pub fn chip_device_free(ptr: *mut c_void);
    pub fn chip_pinned_alloc(size: usize) -> *mut c_void;
    pub fn chip_pinned_free(ptr: *mut c_void);
    pub fn chip_memcpy_h2d(dst_dev: *mut c_void, src_host: *const c_void, size: usize, stream: *mut c_void) -> c_int;
    pub fn chip_memcpy_d2h(dst_host: *mut c_void, src_dev: *const c_void, size: usize, stream: *mut c_void) -> c_int;
    pub fn chip_stream_sync(stream: *mut c_void) -> c_int;

    pub fn chip_decoder_new(format: c_int, opts: *const chip_decoder_opts) -> *mut chip_decoder;
    pub fn chip_decode(d: *mut chip_decoder, input: *const u8, input_len: usize, output: *mut u8, output_len: usize) -> chip_decode_result;
    pub fn chip_decoder_reset(d: *mut chip_decoder) -> *mut chip_decoder;
    pub fn chip_decoder_free(d: *mut chip_decoder);
    pub fn chip_decoder_strerror(format: c_int, code: i32) -> *const c_char;

    pub fn chip_encoder_new(opts: *const chip_encoder_opts) -> *mut chip_encoder;
    pub fn chip_encode(e: *mut chip_encoder, input: *const u8, input_len: usize, output: *mut u8, output_len: usize, op: c_int) -> chip_encode_result;
    pub fn chip_encoder_reset(e: *mut chip_encoder) -> *mut chip_encoder;
    pub fn chip_encoder_free(e: *mut chip_encoder);
}
