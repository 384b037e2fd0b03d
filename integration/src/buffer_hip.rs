// Appended to src/buffer.rs (feature "hip"): the pinned-host and device buffer types, with `Buffer<N>`'s cursor API.
// The C++ twins (compu_amd/host/compu.hpp: PinnedBuffer, DeviceBuffer) are what tests/cpp/test_reference.cpp runs.

///Page-locked host buffer: `Buffer<N>` whose storage the GPU can DMA to and from.
///
///`Buffer::decode` / `Buffer::encode` (src/decoder/mod.rs:507-531, src/encoder/mod.rs:395-412) get twin impls over it;
///they only use `cursor`, `spare_capacity_mut` and `data`.
#[cfg(feature = "hip")]
pub struct PinnedBuffer {
    buffer: core::ptr::NonNull<u8>,
    capacity: usize,
    pub(crate) cursor: usize,
}

#[cfg(feature = "hip")]
impl PinnedBuffer {
    ///Creates new instance, `None` when page-locked memory is not available
    pub fn new(capacity: usize) -> Option<Self> {
        debug_assert!(capacity >= 128, "Buffer less than 128 bytes makes no sense");
        core::ptr::NonNull::new(crate::mem::hip_pinned_alloc(capacity)).map(|buffer| Self { buffer, capacity, cursor: 0 })
    }

    #[inline(always)]
    ///Returns unconsumed data
    pub fn data(&self) -> &[u8] {
        unsafe { core::slice::from_raw_parts(self.buffer.as_ptr(), self.cursor) }
    }

    #[inline(always)]
    ///Marks internal buffer as consumed fully
    pub fn consume(&mut self) {
        self.cursor = 0;
    }

    #[inline(always)]
    ///Returns spare capacity in buffer
    pub fn spare_capacity_mut(&mut self) -> &mut [core::mem::MaybeUninit<u8>] {
        unsafe { core::slice::from_raw_parts_mut(self.buffer.as_ptr().add(self.cursor) as _, self.capacity - self.cursor) }
    }
}

#[cfg(feature = "hip")]
impl Drop for PinnedBuffer {
    fn drop(&mut self) {
        unsafe { crate::hip_sys::chip_pinned_free(self.buffer.as_ptr() as _) }
    }
}

///Buffer in the memory of the current GPU.  The batched entry points (`chip_decode_batch`, `chip_encode_batch`) read
///and write device memory; a batch decodes out of one `DeviceBuffer` into the spare capacity of another.
#[cfg(feature = "hip")]
pub struct DeviceBuffer {
    buffer: core::ptr::NonNull<u8>,
    capacity: usize,
    pub(crate) cursor: usize,
}

#[cfg(feature = "hip")]
impl DeviceBuffer {
    ///Creates new instance, `None` when the device allocation fails
    pub fn new(capacity: usize) -> Option<Self> {
        core::ptr::NonNull::new(crate::mem::hip_device_alloc(capacity + 16)).map(|buffer| Self { buffer, capacity, cursor: 0 })
    }

    #[inline(always)]
    ///DEVICE pointer to the written part and its length
    pub fn data(&self) -> (*const u8, usize) {
        (self.buffer.as_ptr(), self.cursor)
    }

    #[inline(always)]
    ///Size of the allocation in bytes
    pub fn capacity(&self) -> usize {
        self.capacity
    }

    #[inline(always)]
    ///DEVICE pointer to the buffer's first byte (for the batched entry points, which take whole buffers plus offsets)
    pub fn as_ptr(&self) -> *const u8 {
        self.buffer.as_ptr()
    }

    #[inline(always)]
    ///Mutable DEVICE pointer to the buffer's first byte
    pub fn as_mut_ptr(&mut self) -> *mut u8 {
        self.buffer.as_ptr()
    }

    #[inline(always)]
    ///Marks internal buffer as consumed fully
    pub fn consume(&mut self) {
        self.cursor = 0;
    }

    #[inline(always)]
    ///DEVICE pointer to the spare capacity and its length
    pub fn spare_capacity_mut(&mut self) -> (*mut u8, usize) {
        (unsafe { self.buffer.as_ptr().add(self.cursor) }, self.capacity - self.cursor)
    }

    ///Appends host bytes (synchronous copy); `false` when they do not fit
    pub fn upload(&mut self, data: &[u8]) -> bool {
        if data.len() > self.capacity - self.cursor {
            return false;
        }
        let ok = unsafe {
            crate::hip_sys::chip_memcpy_h2d(self.buffer.as_ptr().add(self.cursor) as _, data.as_ptr() as _, data.len(), core::ptr::null_mut()) == 0
                && crate::hip_sys::chip_stream_sync(core::ptr::null_mut()) == 0
        };
        if ok {
            self.cursor += data.len();
        }
        ok
    }
}

#[cfg(feature = "hip")]
impl Drop for DeviceBuffer {
    fn drop(&mut self) {
        unsafe { crate::hip_sys::chip_device_free(self.buffer.as_ptr() as _) }
    }
}
