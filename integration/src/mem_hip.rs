// Appended to src/mem.rs (feature "hip").  The CPU backends hand `compu_alloc` / `compu_free_with_state` to the C
// library per stream (z_stream.zalloc / ZSTD_customMem); the hip backend takes the same two functions once, through
// chip_set_allocator, and every decoder / encoder keeps the pair it was created with.  Device memory comes from hipMalloc
// and staging memory from hipHostMalloc (chip_device_alloc / chip_pinned_alloc below): payload never goes through
// the Rust allocator, as with the CPU backends.

#[cfg(feature = "hip")]
pub(crate) fn hip_install_allocator() {
    use core::sync::atomic::{AtomicBool, Ordering};
    static DONE: AtomicBool = AtomicBool::new(false);
    if !DONE.swap(true, Ordering::AcqRel) {
        unsafe extern "C" fn malloc(_: *mut core::ffi::c_void, size: usize) -> *mut core::ffi::c_void {
            compu_malloc_with_state(core::ptr::null_mut(), size) as _
        }
        unsafe extern "C" fn free(_: *mut core::ffi::c_void, ptr: *mut core::ffi::c_void) {
            compu_free_with_state(core::ptr::null_mut(), ptr as _)
        }
        unsafe {
            crate::hip_sys::chip_set_allocator(Some(malloc), Some(free), core::ptr::null_mut());
        }
    }
}

///Device memory of the current GPU (`hipMalloc`); null on failure
#[cfg(feature = "hip")]
#[inline]
pub(crate) fn hip_device_alloc(size: usize) -> *mut u8 {
    unsafe { crate::hip_sys::chip_device_alloc(size) as *mut u8 }
}

///Page-locked host memory (`hipHostMalloc`); null on failure
#[cfg(feature = "hip")]
#[inline]
pub(crate) fn hip_pinned_alloc(size: usize) -> *mut u8 {
    unsafe { crate::hip_sys::chip_pinned_alloc(size) as *mut u8 }
}
