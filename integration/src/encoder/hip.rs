//! `hip` module: zlib / gzip / raw deflate encoding on an MI355X through `libcompu_hip.so`.

extern crate alloc;

use core::ptr;

use super::{Encode, EncodeOp, EncodeStatus, Encoder, Interface, ZlibOptions, ZlibStrategy};
use crate::hip_sys as sys;

static HIP_ZLIB: Interface = Interface {
    drop_fn,
    reset_fn,
    encode_fn,
};

impl Interface {
    #[inline]
    ///Creates encoder with `hip` interface
    ///
    ///Returns `None` if unable to initialize it (no usable GPU, or lack of memory)
    pub fn zlib_hip(opts: ZlibOptions) -> Option<Encoder> {
        crate::mem::hip_install_allocator();
        let opts = sys::chip_encoder_opts {
            mode: opts.mode as _,
            compression: opts.compression as _,
            device: -1,
            strategy: match opts.strategy {
                ZlibStrategy::Default => 0,
                ZlibStrategy::Filtered => 1,
                ZlibStrategy::HuffmanOnly => 2,
                ZlibStrategy::Rle => 3,
                ZlibStrategy::Fixed => 4,
            },
            mem_level: opts.mem_level as _,
        };
        let instance = unsafe { sys::chip_encoder_new(&opts) };
        //like zlib_ng (src/encoder/zlib_ng.rs:84): the options live in the state, the replayed `opts` bytes are unused
        ptr::NonNull::new(instance as *mut u8).map(|instance| HIP_ZLIB.inner_encoder(instance, [0; 2]))
    }
}

unsafe fn encode_fn(state: ptr::NonNull<u8>, input: *const u8, input_remain: usize, output: *mut u8, output_remain: usize, op: EncodeOp) -> Encode {
    let op = match op {
        EncodeOp::Process => 0,
        EncodeOp::Flush => 1,
        EncodeOp::Finish => 2,
    };
    let result = sys::chip_encode(state.as_ptr() as *mut sys::chip_encoder, input, input_remain, output, output_remain, op);
    Encode {
        input_remain: result.input_remain,
        output_remain: result.output_remain,
        status: match result.status {
            0 => EncodeStatus::Continue,
            1 => EncodeStatus::NeedOutput,
            2 => EncodeStatus::Finished,
            _ => EncodeStatus::Error,
        },
    }
}

#[inline]
fn reset_fn(state: ptr::NonNull<u8>, _: [u8; 2]) -> Option<ptr::NonNull<u8>> {
    let result = unsafe { sys::chip_encoder_reset(state.as_ptr() as *mut sys::chip_encoder) };
    ptr::NonNull::new(result as *mut u8)
}

#[inline]
fn drop_fn(state: ptr::NonNull<u8>) {
    unsafe {
        sys::chip_encoder_free(state.as_ptr() as *mut sys::chip_encoder);
    }
}

// ---- the batched path -----------------------------------------------------------------------------------------
//
// Safe face of `chip_encode_batch*`: every unit becomes one complete stream of `opts.mode` (wrapper, deflate body, trailer),
// as `Interface::zlib_hip(opts)` -> `encode(.., EncodeOp::Finish)` -> `reset` would produce it one unit at a time
// (src/encoder/zlib_ng.rs:50-104, src/encoder/mod.rs:334-370).

///One unit of an encode batch: the range to compress and where its stream goes.
#[derive(Clone, Copy, Debug)]
pub struct EncodeUnit {
    ///offset of the unit's first byte in the input buffer
    pub in_off: u64,
    ///its length
    pub in_len: u32,
    ///offset of the unit's output range in the output buffer
    pub out_off: u64,
    ///capacity of that range (`encode_bound` is always enough)
    pub out_cap: u32,
}

fn strategy_tag(strategy: ZlibStrategy) -> core::ffi::c_int {
    match strategy {
        ZlibStrategy::Default => 0,
        ZlibStrategy::Filtered => 1,
        ZlibStrategy::HuffmanOnly => 2,
        ZlibStrategy::Rle => 3,
        ZlibStrategy::Fixed => 4,
    }
}

///Capacity that always holds the stream of `in_len` input bytes in `opts.mode`.
pub fn encode_bound(opts: &ZlibOptions, in_len: usize) -> usize {
    unsafe { sys::chip_encode_bound(opts.mode as _, in_len) }
}

///Compresses `units` of `input` into `output`, both in HOST memory, on `device`.  Returns, per unit, the compressed length and the
///status (`EncodeStatus::Finished`, or `NeedOutput` when `out_cap` was too small).
pub fn encode_batch_host(opts: &ZlibOptions, device: i32, input: &[u8], units: &[EncodeUnit], output: &mut [u8]) -> Result<alloc::vec::Vec<(u32, EncodeStatus)>, i32> {
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
    let mut status = alloc::vec![0i32; n];
    //Default / Filtered go through the plain entry point; the other strategies need the device-resident `_ex` form, which the
    //maintainer reaches through `encode_batch_device`
    if strategy_tag(opts.strategy) > 1 {
        return Err(-101);
    }
    let rc = unsafe {
        sys::chip_encode_batch_host(opts.mode as _, opts.compression as _, n, input.as_ptr() as *const _, in_off.as_ptr(), in_len.as_ptr(),
                                    output.as_mut_ptr() as *mut _, out_off.as_ptr(), out_cap.as_ptr(), out_len.as_mut_ptr(), status.as_mut_ptr(), device, 0)
    };
    if rc != sys::CHIP_OK {
        return Err(rc);
    }
    Ok((0..n)
        .map(|i| {
            (out_len[i], match status[i] {
                2 => EncodeStatus::Finished,
                1 => EncodeStatus::NeedOutput,
                _ => EncodeStatus::Error,
            })
        })
        .collect())
}

///Device-resident encode batch with the full option surface (`chip_encode_batch_ex`): one launch on `stream`.
///
///# Safety
///
///The offset / length arrays are read by the GPU: they must describe ranges inside `input` and `output`, and all buffers must
///stay alive until the stream has been synchronised.
pub unsafe fn encode_batch_device(opts: &ZlibOptions, n: usize, input: &crate::buffer::DeviceBuffer, in_off: &crate::buffer::DeviceBuffer,
                                  in_len: &crate::buffer::DeviceBuffer, output: &mut crate::buffer::DeviceBuffer, out_off: &crate::buffer::DeviceBuffer,
                                  out_cap: &crate::buffer::DeviceBuffer, out_len: &mut crate::buffer::DeviceBuffer, status: &mut crate::buffer::DeviceBuffer,
                                  stream: *mut core::ffi::c_void) -> Result<(), i32> {
    if in_off.capacity() < 8 * n || out_off.capacity() < 8 * n || in_len.capacity() < 4 * n || out_cap.capacity() < 4 * n || out_len.capacity() < 4 * n
        || status.capacity() < 4 * n
    {
        return Err(-101);
    }
    let rc = if strategy_tag(opts.strategy) == 0 {
        sys::chip_encode_batch(opts.mode as _, opts.compression as _, n, input.as_ptr() as *const _, in_off.as_ptr() as *const u64,
                               in_len.as_ptr() as *const u32, output.as_mut_ptr() as *mut _, out_off.as_ptr() as *const u64, out_cap.as_ptr() as *const u32,
                               out_len.as_mut_ptr() as *mut u32, status.as_mut_ptr() as *mut i32, stream)
    } else {
        sys::chip_encode_batch_ex(opts.mode as _, opts.compression as _, strategy_tag(opts.strategy), n, input.as_ptr() as *const _,
                                  in_off.as_ptr() as *const u64, in_len.as_ptr() as *const u32, output.as_mut_ptr() as *mut _, out_off.as_ptr() as *const u64,
                                  out_cap.as_ptr() as *const u32, out_len.as_mut_ptr() as *mut u32, status.as_mut_ptr() as *mut i32, stream)
    };
    if rc == sys::CHIP_OK { Ok(()) } else { Err(rc) }
}
