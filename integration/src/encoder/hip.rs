//! `hip` module: zlib / gzip / raw deflate encoding on an MI355X through `libcompu_hip.so`.

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
