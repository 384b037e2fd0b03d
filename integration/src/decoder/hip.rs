//! `hip` interface implementation: zlib / gzip / raw deflate and zstd decoding on an MI355X through `libcompu_hip.so`.
//!
//! Same shape as `zlib_ng.rs`: one static vtable per format family, the state pointer is the backend's opaque
//! decoder object, created by the constructor and freed exactly once by `drop_fn`.

extern crate alloc;

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

// ---- the batched hot path ------------------------------------------------------------------------------------
//
// compu's `Decoder` decodes one stream per call.  The MI355X backend earns its keep on BATCHES of independent units (one
// wavefront per unit, one launch per batch): these free functions are the safe face of `chip_decode_batch*`.  They sit
// beside the vtable of src/decoder/mod.rs:160-166 and do, per unit, what the loop
// `Interface::zlib_ng(mode)` -> `decode` -> `reset` (src/decoder/zlib_ng.rs:61-108) does on the CPU.

///One unit of a batch: where its compressed bytes lie in the input buffer and where its output goes.
#[derive(Clone, Copy, Debug)]
pub struct BatchUnit {
    ///offset of the unit's first compressed byte in the input buffer
    pub in_off: u64,
    ///compressed length
    pub in_len: u32,
    ///offset of the unit's output range in the output buffer
    pub out_off: u64,
    ///capacity of that range
    pub out_cap: u32,
}

///What the backend reports per unit.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct BatchResult {
    ///bytes written
    pub out_len: u32,
    ///input bytes consumed
    pub in_used: u32,
    ///`Ok(status)` or the codec's error code, as `Decode::status`
    pub status: Result<DecodeStatus, DecodeError>,
}

///Batch format: one of `ZlibMode`'s window-bits values, zstd, or per-unit routing by `Detection::detect`.
///
///(`ZlibMode` derives only `Copy` and `Clone`, src/decoder/zlib_common.rs:1, so `Debug` is written by hand.)
#[derive(Clone, Copy)]
pub enum BatchFormat {
    ///raw deflate / zlib / gzip / zlib-or-gzip, as `Interface::zlib_hip(mode)`
    Zlib(ZlibMode),
    ///zstd frames
    Zstd,
    ///gzip, zlib and zstd units mixed: each unit goes where `Detection::detect` sends it (src/decoder/mod.rs:28-114)
    Detect,
}

impl core::fmt::Debug for BatchFormat {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        match self {
            BatchFormat::Zlib(mode) => write!(f, "Zlib({})", mode.max_bits()),
            BatchFormat::Zstd => f.write_str("Zstd"),
            BatchFormat::Detect => f.write_str("Detect"),
        }
    }
}

impl BatchFormat {
    fn tag(self) -> core::ffi::c_int {
        match self {
            BatchFormat::Zlib(mode) => mode.max_bits() as _,
            BatchFormat::Zstd => sys::CHIP_FMT_ZSTD,
            BatchFormat::Detect => sys::CHIP_FMT_DETECT,
        }
    }
}

fn map_status(status: i32) -> Result<DecodeStatus, DecodeError> {
    match status {
        0 => Ok(DecodeStatus::NeedInput),
        1 => Ok(DecodeStatus::NeedOutput),
        2 => Ok(DecodeStatus::Finished),
        code => Err(DecodeError(code)),
    }
}

///Decodes `units` of `input` into `output`, both in HOST memory (pinned memory -- `crate::buffer::PinnedBuffer` -- lets the copies
///overlap the kernels), on `device`.
///
///Returns the backend's error code when the launch itself failed; per-unit outcomes are in the returned vector.
pub fn decode_batch_host(format: BatchFormat, device: i32, input: &[u8], units: &[BatchUnit], output: &mut [u8]) -> Result<alloc::vec::Vec<BatchResult>, i32> {
    let n = units.len();
    for unit in units {
        //the backend trusts offsets and lengths: check them here, once, on the safe side of the boundary
        let in_end = unit.in_off.checked_add(unit.in_len as u64).ok_or(-101)?;
        let out_end = unit.out_off.checked_add(unit.out_cap as u64).ok_or(-101)?;
        if in_end > input.len() as u64 || out_end > output.len() as u64 {
            return Err(-101);
        }
    }
    let in_off: alloc::vec::Vec<u64> = units.iter().map(|u| u.in_off).collect();
    let in_len: alloc::vec::Vec<u32> = units.iter().map(|u| u.in_len).collect();
    let out_off: alloc::vec::Vec<u64> = units.iter().map(|u| u.out_off).collect();
    let out_cap: alloc::vec::Vec<u32> = units.iter().map(|u| u.out_cap).collect();
    let mut out_len = alloc::vec![0u32; n];
    let mut in_used = alloc::vec![0u32; n];
    let mut status = alloc::vec![0i32; n];
    let rc = unsafe {
        sys::chip_decode_batch_host(format.tag(), n, input.as_ptr() as *const _, in_off.as_ptr(), in_len.as_ptr(), output.as_mut_ptr() as *mut _,
                                    out_off.as_ptr(), out_cap.as_ptr(), out_len.as_mut_ptr(), in_used.as_mut_ptr(), status.as_mut_ptr(), device, 0)
    };
    if rc != sys::CHIP_OK {
        return Err(rc);
    }
    Ok((0..n).map(|i| BatchResult { out_len: out_len[i], in_used: in_used[i], status: map_status(status[i]) }).collect())
}

///The same over every visible GPU of the node (or `devices`): the units are partitioned on the host, no collective.
pub fn decode_batch_multi(format: BatchFormat, devices: &[i32], input: &[u8], units: &[BatchUnit], output: &mut [u8]) -> Result<alloc::vec::Vec<BatchResult>, i32> {
    let n = units.len();
    for unit in units {
        let in_end = unit.in_off.checked_add(unit.in_len as u64).ok_or(-101)?;
        let out_end = unit.out_off.checked_add(unit.out_cap as u64).ok_or(-101)?;
        if in_end > input.len() as u64 || out_end > output.len() as u64 {
            return Err(-101);
        }
    }
    let in_off: alloc::vec::Vec<u64> = units.iter().map(|u| u.in_off).collect();
    let in_len: alloc::vec::Vec<u32> = units.iter().map(|u| u.in_len).collect();
    let out_off: alloc::vec::Vec<u64> = units.iter().map(|u| u.out_off).collect();
    let out_cap: alloc::vec::Vec<u32> = units.iter().map(|u| u.out_cap).collect();
    let mut out_len = alloc::vec![0u32; n];
    let mut in_used = alloc::vec![0u32; n];
    let mut status = alloc::vec![0i32; n];
    let rc = unsafe {
        sys::chip_decode_batch_multi(format.tag(), n, input.as_ptr() as *const _, in_off.as_ptr(), in_len.as_ptr(), output.as_mut_ptr() as *mut _,
                                     out_off.as_ptr(), out_cap.as_ptr(), out_len.as_mut_ptr(), in_used.as_mut_ptr(), status.as_mut_ptr(),
                                     if devices.is_empty() { ptr::null() } else { devices.as_ptr() }, devices.len() as _, 0)
    };
    if rc != sys::CHIP_OK {
        return Err(rc);
    }
    Ok((0..n).map(|i| BatchResult { out_len: out_len[i], in_used: in_used[i], status: map_status(status[i]) }).collect())
}

///Device-resident batch: everything -- data, offsets, results -- already lies in `DeviceBuffer`s (src/buffer.rs grows them, see
///`buffer_hip.rs`); the call only enqueues one launch on `stream` (null = default stream).
///
///# Safety
///
///The offset / length arrays are read by the GPU: they must describe ranges inside `input` and `output`, and all buffers must
///stay alive until the stream has been synchronised.
///
///`compu_status`: report every unit's status exactly as this crate's `decode_fn` would have (`CHIP_F_COMPU_STATUS`); `false` names the
///limit that was hit (see `include/compu_hip.h`).
pub unsafe fn decode_batch_device(format: BatchFormat, compu_status: bool, n: usize, input: &crate::buffer::DeviceBuffer, in_off: &crate::buffer::DeviceBuffer,
                                  in_len: &crate::buffer::DeviceBuffer, output: &mut crate::buffer::DeviceBuffer, out_off: &crate::buffer::DeviceBuffer,
                                  out_cap: &crate::buffer::DeviceBuffer, out_len: &mut crate::buffer::DeviceBuffer, in_used: &mut crate::buffer::DeviceBuffer,
                                  status: &mut crate::buffer::DeviceBuffer, stream: *mut core::ffi::c_void) -> Result<(), i32> {
    if in_off.capacity() < 8 * n || out_off.capacity() < 8 * n || in_len.capacity() < 4 * n || out_cap.capacity() < 4 * n || out_len.capacity() < 4 * n
        || in_used.capacity() < 4 * n || status.capacity() < 4 * n
    {
        return Err(-101);
    }
    let flags = if compu_status { sys::CHIP_F_COMPU_STATUS } else { 0 };
    let rc = sys::chip_decode_batch_ex(format.tag(), flags, n, input.as_ptr() as *const _, in_off.as_ptr() as *const u64, in_len.as_ptr() as *const u32,
                                       output.as_mut_ptr() as *mut _, out_off.as_ptr() as *const u64, out_cap.as_ptr() as *const u32,
                                       out_len.as_mut_ptr() as *mut u32, in_used.as_mut_ptr() as *mut u32, status.as_mut_ptr() as *mut i32, stream);
    if rc == sys::CHIP_OK { Ok(()) } else { Err(rc) }
}

///`Detection::detect` for every unit of a device-resident batch (`kind[i]` gets the backend's CHIP_DETECT_* value).
///
///# Safety
///
///As `decode_batch_device`.
pub unsafe fn detect_batch_device(n: usize, input: &crate::buffer::DeviceBuffer, in_off: &crate::buffer::DeviceBuffer, in_len: &crate::buffer::DeviceBuffer,
                                  kind: &mut crate::buffer::DeviceBuffer, stream: *mut core::ffi::c_void) -> Result<(), i32> {
    if in_off.capacity() < 8 * n || in_len.capacity() < 4 * n || kind.capacity() < 4 * n {
        return Err(-101);
    }
    let rc = sys::chip_detect_batch(n, input.as_ptr() as *const _, in_off.as_ptr() as *const u64, in_len.as_ptr() as *const u32,
                                    kind.as_mut_ptr() as *mut i32, stream);
    if rc == sys::CHIP_OK { Ok(()) } else { Err(rc) }
}

///The partition `decode_batch_multi` uses: `parts + 1` cut points, worker `w` owns units `cuts[w]..cuts[w + 1]`.
pub fn partition_units(in_len: &[u32], out_cap: &[u32], parts: usize) -> Option<alloc::vec::Vec<usize>> {
    if in_len.len() != out_cap.len() || parts == 0 {
        return None;
    }
    let mut cuts = alloc::vec![0usize; parts + 1];
    let rc = unsafe { sys::chip_partition_units(in_len.len(), in_len.as_ptr(), out_cap.as_ptr(), parts as _, cuts.as_mut_ptr()) };
    if rc == sys::CHIP_OK { Some(cuts) } else { None }
}
