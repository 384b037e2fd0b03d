// C ABI of libcompu_hip.so (include/compu_hip.h): device/buffer management, the batched entry
// points, and the streaming Decoder/Encoder objects that mirror compu's vtables
// (src/decoder/mod.rs:160-166, src/encoder/mod.rs:52-57) on top of the batched GPU kernels.
//
// The streaming objects keep compu's call contract (NeedInput / NeedOutput / Finished, borrowed
// host buffers) but do all codec work on the GPU: input is accumulated in pinned host memory,
// the whole stream seen so far is decoded by the batch kernel (a batch of one), and decoded bytes
// are handed out of a device buffer as the caller provides room -- the same shape as brotli's
// internally buffered decoder that tests/decoder.rs:38-39 already allows for.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "chip_internal.h"

using namespace chip;

namespace {

chip_malloc_fn g_malloc = nullptr;
chip_free_fn g_free = nullptr;
void *g_opaque = nullptr;

void *host_alloc(size_t n) { return g_malloc ? g_malloc(g_opaque, n) : malloc(n); }
void host_free(void *p)
{
    if (!p) return;
    if (g_free) g_free(g_opaque, p);
    else free(p);
}

bool device_ok()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

// small device+pinned block holding the per-unit descriptor and result words of a batch of one
struct Meta {
    uint64_t in_off, out_off;
    uint32_t in_len, out_cap, out_len, in_used;
    int32_t status;
    uint32_t resume[3];  // inflate streaming: see BatchArgs::resume
};

}  // namespace

extern "C" {

int chip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int chip_set_device(int device) { return hipSetDevice(device) == hipSuccess ? CHIP_OK : CHIP_E_NO_DEVICE; }

const char *chip_version(void) { return "compu-hip 0.1 (gfx950)"; }

void chip_set_allocator(chip_malloc_fn malloc_fn, chip_free_fn free_fn, void *opaque)
{
    g_malloc = malloc_fn;
    g_free = free_fn;
    g_opaque = opaque;
}

void *chip_device_alloc(size_t size)
{
    void *p = nullptr;
    if (hipMalloc(&p, size ? size : 4) != hipSuccess) return nullptr;
    return p;
}
void chip_device_free(void *ptr)
{
    if (ptr) (void)hipFree(ptr);
}
void *chip_pinned_alloc(size_t size)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, size ? size : 4, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void chip_pinned_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}
int chip_memcpy_h2d(void *dst, const void *src, size_t size, void *stream)
{
    return hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess ? CHIP_OK : CHIP_E_LAUNCH;
}
int chip_memcpy_d2h(void *dst, const void *src, size_t size, void *stream)
{
    return hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? CHIP_OK : CHIP_E_LAUNCH;
}
int chip_stream_sync(void *stream) { return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? CHIP_OK : CHIP_E_LAUNCH; }
int chip_trim(void) { return chip::release_inflate_scratch() == hipSuccess ? CHIP_OK : CHIP_E_LAUNCH; }

// ---- batched decode ------------------------------------------------------------------------

int chip_decode_batch(int format, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                      void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                      uint32_t *in_used, int32_t *status, void *stream)
{
    if (n == 0) return CHIP_OK;
    if (n > 0x7fffffffull || !in_base || !in_off || !in_len || !out_base || !out_off || !out_cap || !out_len || !in_used ||
        !status || ((uintptr_t)in_base & 3u))
        return CHIP_E_INVALID;
    if (!device_ok()) return CHIP_E_NO_DEVICE;
    BatchArgs a;
    a.in_base = (const uint8_t *)in_base;
    a.in_off = in_off;
    a.in_len = in_len;
    a.out_base = (uint8_t *)out_base;
    a.out_off = out_off;
    a.out_cap = out_cap;
    a.out_len = out_len;
    a.in_used = in_used;
    a.status = status;
    a.n = (uint32_t)n;
    a.format = format;
    a.stats = nullptr;
    a.resume = nullptr;
#ifdef CHIP_STATS
    a.stats = (unsigned long long *)getenv("CHIP_STATS_PTR") ? (unsigned long long *)strtoull(getenv("CHIP_STATS_PTR"), nullptr, 0) : nullptr;
#endif
    hipError_t e;
    switch (format) {
    case CHIP_FMT_DEFLATE:
    case CHIP_FMT_ZLIB:
    case CHIP_FMT_GZIP:
    case CHIP_FMT_AUTO: e = launch_inflate(a, (hipStream_t)stream); break;
    case CHIP_FMT_ZSTD: e = launch_zstd_decode(a, 0, (hipStream_t)stream); break;
    case CHIP_FMT_DETECT:  // both kernels see every unit; each takes the ones Detection::detect assigns to it
        e = launch_inflate(a, (hipStream_t)stream);
        if (e == hipSuccess) e = launch_zstd_decode(a, 0, (hipStream_t)stream);
        break;
    default: return CHIP_E_INVALID;
    }
    return e == hipSuccess ? CHIP_OK : CHIP_E_LAUNCH;
}

}  // extern "C"

namespace {
// Host-memory batches: slices of consecutive units alternate between two streams (H2D -> kernel -> D2H each).
// `launch` enqueues the device batch call for one slice; in_used may be null (encode has no such result).
template <class Launch>
int host_pipeline(size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len, void *out_base,
                  const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, uint32_t *in_used, int32_t *status,
                  int device, size_t slice_bytes, Launch launch)
{
    if (n == 0) return CHIP_OK;
    if (n > 0x7fffffffull || !in_base || !in_off || !in_len || !out_base || !out_off || !out_cap || !out_len || !status)
        return CHIP_E_INVALID;
    if (!device_ok()) return CHIP_E_NO_DEVICE;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return CHIP_E_INVALID;
    if (slice_bytes == 0) slice_bytes = (size_t)256 << 20;
    struct Lane {  // one of the two pipelines
        hipStream_t stream = nullptr;
        uint8_t *d_in = nullptr, *d_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        uint8_t *d_arr = nullptr;  // per-unit arrays of the slice, device side
        uint8_t *h_arr = nullptr;  // the same, pinned host side
        size_t arr_units = 0;
        size_t i0 = 0, i1 = 0;     // slice in flight (i1 == i0: none)
        uint64_t out_lo = 0;
    } lanes[2];
    int rc = CHIP_OK;
    // per unit: in_off u64, out_off u64, in_len u32, out_cap u32, out_len u32, in_used u32, status i32 = 36 bytes
    auto arr_bytes = [](size_t units) { return units * 36 + 64; };
    auto reserve = [&](Lane &ln, size_t in_need, size_t out_need, size_t units) -> bool {
        if (in_need > ln.in_cap) {
            chip_device_free(ln.d_in);
            ln.in_cap = in_need + (in_need >> 2) + 4096;
            if (!(ln.d_in = (uint8_t *)chip_device_alloc(ln.in_cap))) return false;
        }
        if (out_need > ln.out_cap) {
            chip_device_free(ln.d_out);
            ln.out_cap = out_need + (out_need >> 2) + 4096;
            if (!(ln.d_out = (uint8_t *)chip_device_alloc(ln.out_cap))) return false;
        }
        if (units > ln.arr_units) {
            chip_device_free(ln.d_arr);
            chip_pinned_free(ln.h_arr);
            ln.arr_units = units + (units >> 2) + 64;
            ln.d_arr = (uint8_t *)chip_device_alloc(arr_bytes(ln.arr_units));
            ln.h_arr = (uint8_t *)chip_pinned_alloc(arr_bytes(ln.arr_units));
            if (!ln.d_arr || !ln.h_arr) return false;
        }
        return true;
    };
    // results of the slice a lane has in flight -> caller's arrays (after its stream has drained)
    auto collect = [&](Lane &ln) -> bool {
        if (ln.i1 == ln.i0) return true;
        if (hipStreamSynchronize(ln.stream) != hipSuccess) return false;
        const size_t m = ln.i1 - ln.i0;
        const uint32_t *r = (const uint32_t *)(ln.h_arr + m * 24);
        memcpy(out_len + ln.i0, r, m * 4);
        if (in_used) memcpy(in_used + ln.i0, r + m, m * 4);
        memcpy(status + ln.i0, r + 2 * m, m * 4);
        ln.i1 = ln.i0;
        return true;
    };
    for (int k = 0; k < 2 && rc == CHIP_OK; k++)
        if (hipStreamCreateWithFlags(&lanes[k].stream, hipStreamNonBlocking) != hipSuccess) rc = CHIP_E_LAUNCH;
    size_t i = 0;
    int turn = 0;
    while (rc == CHIP_OK && i < n) {
        // next slice: consecutive units until the byte budget is reached
        size_t j = i;
        uint64_t in_lo = ~0ull, in_hi = 0, o_lo = ~0ull, o_hi = 0, budget = 0;
        while (j < n) {
            const uint64_t a0 = in_off[j], a1 = a0 + in_len[j], b0 = out_off[j], b1 = b0 + out_cap[j];
            if (j > i && budget + in_len[j] + out_cap[j] > slice_bytes) break;
            budget += (uint64_t)in_len[j] + out_cap[j];
            in_lo = a0 < in_lo ? a0 : in_lo;
            in_hi = a1 > in_hi ? a1 : in_hi;
            o_lo = b0 < o_lo ? b0 : o_lo;
            o_hi = b1 > o_hi ? b1 : o_hi;
            j++;
        }
        Lane &ln = lanes[turn];
        turn ^= 1;
        if (!collect(ln)) {
            rc = CHIP_E_LAUNCH;
            break;
        }
        const uint64_t in_base_lo = in_lo & ~3ull;  // keep the units' alignment relative to a dword-aligned device base
        const size_t in_span = (size_t)(in_hi - in_base_lo), out_span = (size_t)(o_hi - o_lo), m = j - i;
        if (!reserve(ln, ((in_span + 3) & ~(size_t)3) + 16, out_span + 16, m)) {
            rc = CHIP_E_NOMEM;
            break;
        }
        // per-unit arrays rebased to the slice
        uint64_t *h_in_off = (uint64_t *)ln.h_arr, *h_out_off = h_in_off + m;
        uint32_t *h_in_len = (uint32_t *)(h_out_off + m), *h_out_cap = h_in_len + m;
        for (size_t u = 0; u < m; u++) {
            h_in_off[u] = in_off[i + u] - in_base_lo;
            h_out_off[u] = out_off[i + u] - o_lo;
            h_in_len[u] = in_len[i + u];
            h_out_cap[u] = out_cap[i + u];
        }
        uint64_t *d_in_off = (uint64_t *)ln.d_arr, *d_out_off = d_in_off + m;
        uint32_t *d_in_len = (uint32_t *)(d_out_off + m), *d_out_cap = d_in_len + m, *d_res = d_out_cap + m;
        bool ok = hipMemcpyAsync(ln.d_arr, ln.h_arr, m * 24, hipMemcpyHostToDevice, ln.stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(ln.d_in, (const uint8_t *)in_base + in_base_lo, in_span, hipMemcpyHostToDevice, ln.stream) == hipSuccess;
        ok = ok && launch(m, ln.d_in, d_in_off, d_in_len, ln.d_out, d_out_off, d_out_cap, d_res, d_res + m, (int32_t *)(d_res + 2 * m), ln.stream) == CHIP_OK;
        ok = ok && hipMemcpyAsync((uint8_t *)out_base + o_lo, ln.d_out, out_span, hipMemcpyDeviceToHost, ln.stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(ln.h_arr + m * 24, d_res, m * 12, hipMemcpyDeviceToHost, ln.stream) == hipSuccess;
        if (!ok) {
            rc = CHIP_E_LAUNCH;
            break;
        }
        ln.i0 = i;
        ln.i1 = j;
        i = j;
    }
    for (int k = 0; k < 2; k++) {
        Lane &ln = lanes[k];
        if (ln.stream) {
            if (rc == CHIP_OK) {
                if (!collect(ln)) rc = CHIP_E_LAUNCH;
            } else {
                (void)hipStreamSynchronize(ln.stream);
            }
        }
    }
    for (int k = 0; k < 2; k++) {
        Lane &ln = lanes[k];
        if (ln.stream) {
            chip::release_inflate_scratch_of(ln.stream);  // the stream's token scratch goes with it
            (void)hipStreamDestroy(ln.stream);
        }
        chip_device_free(ln.d_in);
        chip_device_free(ln.d_out);
        chip_device_free(ln.d_arr);
        chip_pinned_free(ln.h_arr);
    }
    return rc;
}
}  // namespace

extern "C" {

int chip_decode_batch_host(int format, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                           void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                           uint32_t *in_used, int32_t *status, int device, size_t slice_bytes)
{
    if (!in_used) return CHIP_E_INVALID;
    return host_pipeline(n, in_base, in_off, in_len, out_base, out_off, out_cap, out_len, in_used, status, device, slice_bytes,
                         [=](size_t m, const void *di, const uint64_t *dio, const uint32_t *dil, void *dout, const uint64_t *doo,
                             const uint32_t *doc, uint32_t *dol, uint32_t *diu, int32_t *dst, void *st) {
                             return chip_decode_batch(format, m, di, dio, dil, dout, doo, doc, dol, diu, dst, st);
                         });
}

int chip_encode_batch_host(int format, int level, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                           void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, int32_t *status,
                           int device, size_t slice_bytes)
{
    return host_pipeline(n, in_base, in_off, in_len, out_base, out_off, out_cap, out_len, nullptr, status, device, slice_bytes,
                         [=](size_t m, const void *di, const uint64_t *dio, const uint32_t *dil, void *dout, const uint64_t *doo,
                             const uint32_t *doc, uint32_t *dol, uint32_t *, int32_t *dst, void *st) {
                             return chip_encode_batch(format, level, m, di, dio, dil, dout, doo, doc, dol, dst, st);
                         });
}

// Detection::detect, src/decoder/mod.rs:28-114 (including the fall-through of the 0x68 arm,
// src/decoder/mod.rs:80-82, which makes `68 xx` zlib headers come out as Unknown).
int chip_detect(const uint8_t *b, size_t len)
{
    if (len < 2) return CHIP_DETECT_NONE;
    if (b[0] == 0x1f && b[1] == 0x8b) return CHIP_DETECT_GZIP;
    if ((((unsigned)b[0] << 8) | b[1]) % 31 == 0) {
        static const uint8_t flg[8][4] = {{0x1d, 0x5b, 0x99, 0xd7}, {0x19, 0x57, 0x95, 0xd3}, {0x15, 0x53, 0x91, 0xcf},
                                          {0x11, 0x4f, 0x8d, 0xcb}, {0x0d, 0x4b, 0x89, 0xc7}, {0x09, 0x47, 0x85, 0xc3},
                                          {0x05, 0x43, 0x81, 0xde}, {0x01, 0x5e, 0x9c, 0xda}};
        if ((b[0] & 0x8f) == 0x08 && b[0] != 0x68) {
            const uint8_t *row = flg[b[0] >> 4];
            for (int i = 0; i < 4; i++)
                if (b[1] == row[i]) return CHIP_DETECT_ZLIB;
        }
    }
    if (len < 4) return CHIP_DETECT_NONE;
    if (b[0] == 0x28 && b[1] == 0xb5 && b[2] == 0x2f && b[3] == 0xfd) return CHIP_DETECT_ZSTD;
    return CHIP_DETECT_UNKNOWN;
}

int chip_detect_batch(size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len, int32_t *kind,
                      void *stream)
{
    if (n == 0) return CHIP_OK;
    if (!in_base || !in_off || !in_len || !kind) return CHIP_E_INVALID;
    if (!device_ok()) return CHIP_E_NO_DEVICE;
    return launch_detect(n, (const uint8_t *)in_base, in_off, in_len, kind, (hipStream_t)stream) == hipSuccess ? CHIP_OK
                                                                                                               : CHIP_E_LAUNCH;
}

// ---- error strings -------------------------------------------------------------------------

const char *chip_decoder_strerror(int format, int32_t code)
{
    if (format == CHIP_FMT_ZSTD) {
        // ZSTD_getErrorName(code as usize), src/decoder/zstd.rs:159-164
        switch (code < 0 ? -code : code) {
        case 0: return "No error detected";
        case 1: return "Error (generic)";
        case 10: return "Unknown frame descriptor";
        case 12: return "Version not supported";
        case 14: return "Unsupported frame parameter";
        case 16: return "Frame requires too much memory for decoding";
        case 20: return "Data corruption detected";
        case 22: return "Restored data doesn't match checksum";
        case 30: return "Dictionary is corrupted";
        case 32: return "Dictionary mismatch";
        case 64: return "Allocation error : not enough memory";
        case 70: return "Destination buffer is too small";
        case 72: return "Src size is incorrect";
        default: return "Unspecified error code";
        }
    }
    // zError, src/decoder/zlib_ng.rs:118-123
    static const char *const tab[] = {"need dictionary", "stream end", "", "file error", "stream error",
                                      "data error", "insufficient memory", "buffer error", "incompatible version", ""};
    int idx = 2 - code;
    if (idx < 0 || idx > 9) idx = 9;
    return tab[idx];
}

}  // extern "C"

// ---- streaming decoder -----------------------------------------------------------------------

struct chip_decoder {
    int format;
    int device;
    int window_log_max;
    hipStream_t stream;
    uint8_t *h_in;   // pinned: every input byte of the current stream
    size_t h_in_cap, h_in_len;
    uint8_t *d_in;
    size_t d_in_cap;
    uint8_t *d_out;
    size_t d_out_cap;
    Meta *d_meta;
    Meta *h_meta;    // pinned
    bool decoded;    // kernel results below describe h_in[0..h_in_len)
    uint32_t k_out_len, k_in_used;
    int32_t k_status;
    size_t delivered;  // decoded bytes already handed to the caller
    bool done;
    size_t d_in_len;      // input bytes already on the device (the stream only grows)
    uint32_t resume[3];   // inflate: last block boundary the kernel reached (BatchArgs::resume); zeros = from the start
    uint32_t *d_zres;     // zstd: device blob with the kernel's block checkpoint (header + decode tables), see zstd.hip
};

namespace {

bool dec_reserve_in(chip_decoder *d, size_t need)
{
    if (need <= d->h_in_cap) return true;
    size_t cap = d->h_in_cap ? d->h_in_cap : 65536;
    while (cap < need) cap *= 2;
    uint8_t *p = (uint8_t *)chip_pinned_alloc(cap);
    if (!p) return false;
    if (d->h_in_len) memcpy(p, d->h_in, d->h_in_len);
    chip_pinned_free(d->h_in);
    d->h_in = p;
    d->h_in_cap = cap;
    return true;
}

// Decode what has accumulated.  Input is appended to the device copy (only the new bytes cross the link), an inflate
// stream continues from the last block boundary an earlier call reached (the output so far stays on the device: it is the
// window and what the trailer checksum covers), so feeding a long stream in pieces costs O(stream), not O(stream^2).
// zstd frames are decoded from their start each time.  The device output grows (contents kept) while it is the limit.
bool dec_run(chip_decoder *d)
{
    if (hipSetDevice(d->device) != hipSuccess) return false;
    const size_t in_len = d->h_in_len;
    const size_t need_in = ((in_len + 3) & ~(size_t)3) + 16;
    if (need_in > d->d_in_cap) {
        const size_t ncap = need_in * 2 + 4096;
        uint8_t *p = (uint8_t *)chip_device_alloc(ncap);
        if (!p) return false;
        chip_device_free(d->d_in);  // (stream-ordered work on it is complete: every call ends with a synchronise)
        d->d_in = p;
        d->d_in_cap = ncap;
        d->d_in_len = 0;  // upload again from the pinned copy
    }
    if (in_len > d->d_in_len) {
        const size_t from = d->d_in_len & ~(size_t)3;  // keep the copies dword aligned
        if (hipMemcpyAsync(d->d_in + from, d->h_in + from, in_len - from, hipMemcpyHostToDevice, d->stream) != hipSuccess) return false;
        d->d_in_len = in_len;
    }
    const bool inflate = d->format != CHIP_FMT_ZSTD;
    constexpr size_t ZRES_BYTES = (16 + 2312) * 4 + 64;  // zstd.hip: ZRES_HDR + ZSAVE_WORDS
    if (!inflate && !d->d_zres) {
        d->d_zres = (uint32_t *)chip_device_alloc(ZRES_BYTES);
        if (!d->d_zres || hipMemsetAsync(d->d_zres, 0, 64, d->stream) != hipSuccess) return false;
    }
    size_t cap = d->d_out_cap;
    if (cap == 0) cap = in_len * 4 > 65536 ? in_len * 4 : 65536;
    for (;;) {
        if (cap > 0xffffffffull) cap = 0xffffffffull;
        if (cap > d->d_out_cap) {
            uint8_t *p = (uint8_t *)chip_device_alloc(cap);
            if (!p) return false;
            const size_t keep = inflate ? (size_t)d->resume[1] : (size_t)d->k_out_len;  // a stream resumes: its output so far must survive
            if (keep && hipMemcpyAsync(p, d->d_out, keep, hipMemcpyDeviceToDevice, d->stream) != hipSuccess) return false;
            if (keep && hipStreamSynchronize(d->stream) != hipSuccess) return false;
            chip_device_free(d->d_out);
            d->d_out = p;
            d->d_out_cap = cap;
        }
        Meta m = {0, 0, (uint32_t)in_len, (uint32_t)d->d_out_cap, 0, 0, 0, {d->resume[0], d->resume[1], d->resume[2]}};
        *d->h_meta = m;
        if (hipMemcpyAsync(d->d_meta, d->h_meta, sizeof(Meta), hipMemcpyHostToDevice, d->stream) != hipSuccess) return false;
        BatchArgs a;
        a.in_base = d->d_in;
        a.in_off = &d->d_meta->in_off;
        a.in_len = &d->d_meta->in_len;
        a.out_base = d->d_out;
        a.out_off = &d->d_meta->out_off;
        a.out_cap = &d->d_meta->out_cap;
        a.out_len = &d->d_meta->out_len;
        a.in_used = &d->d_meta->in_used;
        a.status = &d->d_meta->status;
        a.n = 1;
        a.format = d->format;
        a.stats = nullptr;
        a.resume = inflate ? d->d_meta->resume : d->d_zres;
        hipError_t e = inflate ? launch_inflate(a, d->stream) : launch_zstd_decode(a, d->window_log_max, d->stream);
        if (e != hipSuccess) return false;
        if (hipMemcpyAsync(d->h_meta, d->d_meta, sizeof(Meta), hipMemcpyDeviceToHost, d->stream) != hipSuccess) return false;
        if (hipStreamSynchronize(d->stream) != hipSuccess) return false;
        if (inflate)
            for (int k = 0; k < 3; k++) d->resume[k] = d->h_meta->resume[k];
        d->k_out_len = d->h_meta->out_len;
        if (d->h_meta->status == CHIP_NEED_OUTPUT && d->d_out_cap < 0xffffffffull) {
            cap = d->d_out_cap * 2;
            continue;
        }
        break;
    }
    d->k_out_len = d->h_meta->out_len;
    d->k_in_used = d->h_meta->in_used;
    d->k_status = d->h_meta->status;
    d->decoded = true;
    return true;
}

void dec_clear(chip_decoder *d)
{
    d->h_in_len = 0;
    d->decoded = false;
    d->k_out_len = d->k_in_used = 0;
    d->k_status = CHIP_NEED_INPUT;
    d->delivered = 0;
    d->done = false;
    d->d_in_len = 0;
    d->resume[0] = d->resume[1] = d->resume[2] = 0;
    if (d->d_zres) {  // the next stream starts from its frame header
        (void)hipMemsetAsync(d->d_zres, 0, 64, d->stream);
        (void)hipStreamSynchronize(d->stream);
    }
}

}  // namespace

extern "C" {

chip_decoder *chip_decoder_new(int format, const chip_decoder_opts *opts)
{
    if (format != CHIP_FMT_DEFLATE && format != CHIP_FMT_ZLIB && format != CHIP_FMT_GZIP && format != CHIP_FMT_AUTO &&
        format != CHIP_FMT_ZSTD)
        return nullptr;
    if (!device_ok()) return nullptr;  // no CPU codec behind this backend
    int device = opts ? opts->device : -1;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    chip_decoder *d = (chip_decoder *)host_alloc(sizeof(chip_decoder));
    if (!d) return nullptr;
    memset(d, 0, sizeof *d);
    d->format = format;
    d->device = device;
    d->window_log_max = opts ? opts->window_log_max : 0;
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
        host_free(d);
        return nullptr;
    }
    d->d_meta = (Meta *)chip_device_alloc(sizeof(Meta));
    d->h_meta = (Meta *)chip_pinned_alloc(sizeof(Meta));
    if (!d->d_meta || !d->h_meta || !dec_reserve_in(d, 65536)) {
        chip_decoder_free(d);
        return nullptr;
    }
    dec_clear(d);
    return d;
}

chip_decode_result chip_decode(chip_decoder *d, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    chip_decode_result r = {in_len, out_len, CHIP_NEED_INPUT, 0};
    if (!d) {
        r.status = -1;
        r.err = -2;  // Z_STREAM_ERROR
        return r;
    }
    if (d->done) {
        // like zlib in its DONE state: Finished again, nothing consumed
        r.status = CHIP_FINISHED;
        return r;
    }
    const bool stream_end_known = d->decoded && (d->k_status == CHIP_FINISHED || d->k_status < 0 || d->k_status == CHIP_NEED_DICT);
    size_t taken = 0;
    if (!stream_end_known && in_len) {
        if (!dec_reserve_in(d, d->h_in_len + in_len)) {
            r.status = -1;
            r.err = -4;  // Z_MEM_ERROR
            return r;
        }
        memcpy(d->h_in + d->h_in_len, in, in_len);
        d->h_in_len += in_len;
        taken = in_len;
        d->decoded = false;
    }
    if (!d->decoded && !dec_run(d)) {
        r.status = -1;
        r.err = -4;
        return r;
    }
    size_t avail = d->k_out_len - d->delivered;
    size_t n = avail < out_len ? avail : out_len;
    if (n) {
        (void)hipSetDevice(d->device);
        if (hipMemcpyAsync(out, d->d_out + d->delivered, n, hipMemcpyDeviceToHost, d->stream) != hipSuccess ||
            hipStreamSynchronize(d->stream) != hipSuccess) {
            r.status = -1;
            r.err = -4;
            return r;
        }
        d->delivered += n;
    }
    r.output_remain = out_len - n;
    // bytes of this call that lie behind the end of the stream go back to the caller
    size_t giveback = 0;
    if (d->k_status == CHIP_FINISHED || d->k_status < 0 || d->k_status == CHIP_NEED_DICT) {
        size_t trailing = d->h_in_len - d->k_in_used;
        giveback = trailing < taken ? trailing : taken;
        d->h_in_len -= giveback;
    }
    r.input_remain = (in_len - taken) + giveback;
    if (d->delivered < d->k_out_len) {
        r.status = CHIP_NEED_OUTPUT;
        return r;
    }
    if (d->k_status == CHIP_FINISHED) {
        d->done = true;
        r.status = CHIP_FINISHED;
    } else if (d->k_status == CHIP_NEED_INPUT) {
        // Z_OK with avail_in == 0 -> NeedInput; a call that made no progress at all is zlib's
        // Z_BUF_ERROR, which compu maps to NeedOutput (src/decoder/mod.rs:476-481)
        r.status = (in_len == 0 && n == 0) ? CHIP_NEED_OUTPUT : CHIP_NEED_INPUT;
    } else {
        r.status = -1;
        r.err = d->k_status == CHIP_NEED_DICT ? Z_NEED_DICT : d->k_status;
    }
    return r;
}

chip_decoder *chip_decoder_reset(chip_decoder *d)
{
    if (d) dec_clear(d);
    return d;
}

void chip_decoder_free(chip_decoder *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream) {
        chip::release_inflate_scratch_of(d->stream);  // the stream's token scratch goes with it
        (void)hipStreamDestroy(d->stream);
    }
    chip_pinned_free(d->h_in);
    chip_pinned_free(d->h_meta);
    chip_device_free(d->d_in);
    chip_device_free(d->d_out);
    chip_device_free(d->d_meta);
    chip_device_free(d->d_zres);
    host_free(d);
}

}  // extern "C"

// ---- batched encode --------------------------------------------------------------------------------

extern "C" {

size_t chip_encode_bound(int format, size_t in_len)
{
    size_t blocks = in_len ? (in_len + 65534) / 65535 : 1;
    size_t wrap = format == CHIP_FMT_GZIP ? 18 : format == CHIP_FMT_ZLIB ? 6 : 0;
    return in_len + 5 * blocks + 5 + wrap;
}

int chip_encode_batch(int format, int level, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                      void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, int32_t *status,
                      void *stream)
{
    if (n == 0) return CHIP_OK;
    if (n > 0x7fffffffull || !in_base || !in_off || !in_len || !out_base || !out_off || !out_cap || !out_len || !status ||
        level < 0 || level > 9 || (format != CHIP_FMT_DEFLATE && format != CHIP_FMT_ZLIB && format != CHIP_FMT_GZIP))
        return CHIP_E_INVALID;
    if (!device_ok()) return CHIP_E_NO_DEVICE;
    BatchArgs a;
    a.in_base = (const uint8_t *)in_base;
    a.in_off = in_off;
    a.in_len = in_len;
    a.out_base = (uint8_t *)out_base;
    a.out_off = out_off;
    a.out_cap = out_cap;
    a.out_len = out_len;
    a.in_used = nullptr;
    a.status = status;
    a.n = (uint32_t)n;
    a.format = format;
    a.stats = nullptr;
    a.resume = nullptr;
    hipError_t e = launch_deflate_l1(a, level, 7u, format == CHIP_FMT_ZLIB ? 1u : 0u, 0, nullptr, (hipStream_t)stream);
    return e == hipSuccess ? CHIP_OK : CHIP_E_LAUNCH;
}

}  // extern "C"

// ---- streaming encoder -----------------------------------------------------------------------------
// Same contract as internal_zlib_impl_encode! (src/encoder/mod.rs:334-370): Process buffers input,
// Flush / Finish compress what is buffered as one byte-aligned deflate segment on the GPU (a sync
// marker or the final-block flag closes it), and compressed bytes are handed out as the caller
// provides room.

struct chip_encoder {
    int mode, level, device;
    hipStream_t stream;
    uint8_t *h_in;  // pinned: input not yet compressed
    size_t h_in_cap, h_in_len;
    uint8_t *d_in, *d_out;
    size_t d_in_cap, d_out_cap;
    uint8_t *h_out;  // pinned: compressed bytes not yet delivered
    size_t h_out_cap, h_out_len, delivered;
    struct EMeta {
        uint64_t in_off, out_off;
        uint32_t in_len, out_cap, out_len;
        int32_t status;
        uint32_t check, pad;
    } *d_meta, *h_meta;
    bool started, finished;
    uint32_t check;
    uint64_t total_in;
};

namespace {

bool enc_reserve(uint8_t **buf, size_t *cap, size_t len, size_t need)
{
    if (need <= *cap) return true;
    size_t c = *cap ? *cap : 65536;
    while (c < need) c *= 2;
    uint8_t *p = (uint8_t *)chip_pinned_alloc(c);
    if (!p) return false;
    if (len) memcpy(p, *buf, len);
    chip_pinned_free(*buf);
    *buf = p;
    *cap = c;
    return true;
}

void enc_clear(chip_encoder *e)
{
    e->h_in_len = 0;
    e->h_out_len = e->delivered = 0;
    e->started = e->finished = false;
    e->check = e->mode == CHIP_FMT_ZLIB ? 1u : 0u;
    e->total_in = 0;
}

// compress the buffered input as one segment and append it to h_out
bool enc_segment(chip_encoder *e, bool final)
{
    if (hipSetDevice(e->device) != hipSuccess) return false;
    const size_t n = e->h_in_len;
    const size_t bound = chip_encode_bound(e->mode, n) + 16;
    if (n + 16 > e->d_in_cap) {
        chip_device_free(e->d_in);
        e->d_in_cap = (n + 16) * 2;
        e->d_in = (uint8_t *)chip_device_alloc(e->d_in_cap);
        if (!e->d_in) return false;
    }
    if (bound > e->d_out_cap) {
        chip_device_free(e->d_out);
        e->d_out_cap = bound * 2;
        e->d_out = (uint8_t *)chip_device_alloc(e->d_out_cap);
        if (!e->d_out) return false;
    }
    if (n && hipMemcpyAsync(e->d_in, e->h_in, n, hipMemcpyHostToDevice, e->stream) != hipSuccess) return false;
    chip_encoder::EMeta m = {0, 0, (uint32_t)n, (uint32_t)bound, 0, 0, 0, 0};
    *e->h_meta = m;
    if (hipMemcpyAsync(e->d_meta, e->h_meta, sizeof m, hipMemcpyHostToDevice, e->stream) != hipSuccess) return false;
    BatchArgs a;
    a.in_base = e->d_in;
    a.in_off = &e->d_meta->in_off;
    a.in_len = &e->d_meta->in_len;
    a.out_base = e->d_out;
    a.out_off = &e->d_meta->out_off;
    a.out_cap = &e->d_meta->out_cap;
    a.out_len = &e->d_meta->out_len;
    a.in_used = nullptr;
    a.status = &e->d_meta->status;
    a.n = 1;
    a.format = e->mode;
    a.stats = nullptr;
    a.resume = nullptr;
    const uint32_t flags = (e->started ? 0u : 1u) | (final ? 2u | 4u : 0u);
    if (launch_deflate_l1(a, e->level, flags, e->check, e->total_in, &e->d_meta->check, e->stream) != hipSuccess) return false;
    if (hipMemcpyAsync(e->h_meta, e->d_meta, sizeof m, hipMemcpyDeviceToHost, e->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    if (e->h_meta->status != CHIP_ENC_FINISHED) return false;
    const size_t got = e->h_meta->out_len;
    if (!enc_reserve(&e->h_out, &e->h_out_cap, e->h_out_len, e->h_out_len + got)) return false;
    if (got && (hipMemcpyAsync(e->h_out + e->h_out_len, e->d_out, got, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
                hipStreamSynchronize(e->stream) != hipSuccess))
        return false;
    e->h_out_len += got;
    e->check = e->h_meta->check;
    e->total_in += n;
    e->h_in_len = 0;
    e->started = true;
    if (final) e->finished = true;
    return true;
}

}  // namespace

extern "C" {

chip_encoder *chip_encoder_new(const chip_encoder_opts *opts)
{
    int mode = opts ? opts->mode : CHIP_FMT_GZIP, level = opts ? opts->compression : 9;
    if ((mode != CHIP_FMT_DEFLATE && mode != CHIP_FMT_ZLIB && mode != CHIP_FMT_GZIP) || level < 0 || level > 9) return nullptr;
    if (!device_ok()) return nullptr;  // no CPU codec behind this backend
    int device = opts ? opts->device : -1;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    chip_encoder *e = (chip_encoder *)host_alloc(sizeof(chip_encoder));
    if (!e) return nullptr;
    memset(e, 0, sizeof *e);
    e->mode = mode;
    e->level = level;
    e->device = device;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        host_free(e);
        return nullptr;
    }
    e->d_meta = (chip_encoder::EMeta *)chip_device_alloc(sizeof(chip_encoder::EMeta));
    e->h_meta = (chip_encoder::EMeta *)chip_pinned_alloc(sizeof(chip_encoder::EMeta));
    if (!e->d_meta || !e->h_meta || !enc_reserve(&e->h_in, &e->h_in_cap, 0, 65536) || !enc_reserve(&e->h_out, &e->h_out_cap, 0, 65536)) {
        chip_encoder_free(e);
        return nullptr;
    }
    enc_clear(e);
    return e;
}

chip_encode_result chip_encode(chip_encoder *e, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int op)
{
    chip_encode_result r = {in_len, out_len, CHIP_ENC_ERROR};
    if (!e || op < CHIP_OP_PROCESS || op > CHIP_OP_FINISH) return r;
    size_t taken = 0;
    if (!e->finished && in_len) {
        if (!enc_reserve(&e->h_in, &e->h_in_cap, e->h_in_len, e->h_in_len + in_len)) return r;
        memcpy(e->h_in + e->h_in_len, in, in_len);
        e->h_in_len += in_len;
        taken = in_len;
    }
    if (!e->finished && (op == CHIP_OP_FLUSH || op == CHIP_OP_FINISH) && (e->h_in_len || op == CHIP_OP_FINISH || !e->started)) {
        if (!enc_segment(e, op == CHIP_OP_FINISH)) return r;
    }
    size_t avail = e->h_out_len - e->delivered, k = avail < out_len ? avail : out_len;
    if (k) memcpy(out, e->h_out + e->delivered, k);
    e->delivered += k;
    if (e->delivered == e->h_out_len) e->h_out_len = e->delivered = 0;
    r.input_remain = in_len - taken;
    r.output_remain = out_len - k;
    // deflate() return code -> EncodeStatus, src/encoder/mod.rs:357-367: with Finish, anything short of
    // Z_STREAM_END is NeedOutput; otherwise Z_OK is Continue and a call without progress (Z_BUF_ERROR) NeedOutput
    if (op == CHIP_OP_FINISH) r.status = (e->finished && e->h_out_len == 0) ? CHIP_ENC_FINISHED : CHIP_ENC_NEED_OUTPUT;
    else r.status = (taken || k) ? CHIP_ENC_CONTINUE : CHIP_ENC_NEED_OUTPUT;
    return r;
}

chip_encoder *chip_encoder_reset(chip_encoder *e)
{
    if (e) enc_clear(e);
    return e;
}

void chip_encoder_free(chip_encoder *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    chip_pinned_free(e->h_in);
    chip_pinned_free(e->h_out);
    chip_pinned_free(e->h_meta);
    chip_device_free(e->d_in);
    chip_device_free(e->d_out);
    chip_device_free(e->d_meta);
    host_free(e);
}

}  // extern "C"
