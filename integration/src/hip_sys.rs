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
///route every unit of a batch by `Detection::detect` (src/decoder/mod.rs:28-114)
pub const CHIP_FMT_DETECT: c_int = 0;

///`chip_decode_batch_ex` flag: report `DecodeStatus` exactly as compu's `decode_fn` would
pub const CHIP_F_COMPU_STATUS: u32 = 1;
///`chip_decode_batch` / `chip_encode_batch` return codes
pub const CHIP_OK: c_int = 0;

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

    pub fn chip_set_device(device: c_int) -> c_int;
    pub fn chip_version() -> *const c_char;
    pub fn chip_trim() -> c_int;

    pub fn chip_decoder_new(format: c_int, opts: *const chip_decoder_opts) -> *mut chip_decoder;
    pub fn chip_decode(d: *mut chip_decoder, input: *const u8, input_len: usize, output: *mut u8, output_len: usize) -> chip_decode_result;
    pub fn chip_decoder_reset(d: *mut chip_decoder) -> *mut chip_decoder;
    pub fn chip_decoder_free(d: *mut chip_decoder);
    pub fn chip_decoder_footprint(d: *const chip_decoder, pinned_bytes: *mut usize, device_bytes: *mut usize);
    pub fn chip_decoder_strerror(format: c_int, code: i32) -> *const c_char;

    // ---- the batched hot path (additive API): n independent units per launch, one wavefront per unit
    pub fn chip_decode_batch(format: c_int, n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32, out_base: *mut c_void,
                             out_off: *const u64, out_cap: *const u32, out_len: *mut u32, in_used: *mut u32, status: *mut i32, stream: *mut c_void) -> c_int;
    ///`flags`: `CHIP_F_COMPU_STATUS` makes `status[i]` compu's own reading of the codec's return code (`src/decoder/mod.rs:475-483`,
    ///`src/decoder/zstd.rs:121-133`) where the default names the cause
    pub fn chip_decode_batch_ex(format: c_int, flags: u32, n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32, out_base: *mut c_void,
                                out_off: *const u64, out_cap: *const u32, out_len: *mut u32, in_used: *mut u32, status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn chip_decode_batch_host(format: c_int, n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32, out_base: *mut c_void,
                                  out_off: *const u64, out_cap: *const u32, out_len: *mut u32, in_used: *mut u32, status: *mut i32, device: c_int,
                                  slice_bytes: usize) -> c_int;
    pub fn chip_decode_batch_multi(format: c_int, n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32, out_base: *mut c_void,
                                   out_off: *const u64, out_cap: *const u32, out_len: *mut u32, in_used: *mut u32, status: *mut i32,
                                   devices: *const c_int, n_devices: c_int, slice_bytes: usize) -> c_int;
    pub fn chip_partition_units(n: usize, in_len: *const u32, out_cap: *const u32, parts: c_int, cuts: *mut usize) -> c_int;
    pub fn chip_detect(bytes: *const u8, len: usize) -> c_int;
    pub fn chip_detect_batch(n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32, kind: *mut i32, stream: *mut c_void) -> c_int;

    pub fn chip_encoder_new(opts: *const chip_encoder_opts) -> *mut chip_encoder;
    pub fn chip_encode(e: *mut chip_encoder, input: *const u8, input_len: usize, output: *mut u8, output_len: usize, op: c_int) -> chip_encode_result;
    pub fn chip_encoder_reset(e: *mut chip_encoder) -> *mut chip_encoder;
    pub fn chip_encoder_free(e: *mut chip_encoder);
    pub fn chip_encode_batch_ex(format: c_int, level: c_int, strategy: c_int, n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32,
                                out_base: *mut c_void, out_off: *const u64, out_cap: *const u32, out_len: *mut u32, status: *mut i32,
                                stream: *mut c_void) -> c_int;
    pub fn chip_encode_batch(format: c_int, level: c_int, n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32,
                             out_base: *mut c_void, out_off: *const u64, out_cap: *const u32, out_len: *mut u32, status: *mut i32,
                             stream: *mut c_void) -> c_int;
    pub fn chip_encode_bound(format: c_int, in_len: usize) -> usize;
    pub fn chip_encode_batch_host(format: c_int, level: c_int, n: usize, in_base: *const c_void, in_off: *const u64, in_len: *const u32,
                                  out_base: *mut c_void, out_off: *const u64, out_cap: *const u32, out_len: *mut u32, status: *mut i32, device: c_int,
                                  slice_bytes: usize) -> c_int;
}
