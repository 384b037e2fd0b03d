// Internal declarations shared by the HIP translation units of libcompu_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/compu_hip.h"

namespace chip {

// Per-unit running state codes used inside kernels (never exported).
constexpr int32_t ST_RUNNING = 0x7fffffff;

// zlib return codes carried in DecodeError (src/decoder/mod.rs:482)
constexpr int32_t Z_NEED_DICT = 2;
constexpr int32_t Z_DATA_ERROR = -3;

// ZSTD_ErrorCode values carried (negated) in DecodeError (src/decoder/zstd.rs:131)
constexpr int32_t ZSTD_E_GENERIC = 1;
constexpr int32_t ZSTD_E_PREFIX_UNKNOWN = 10;
constexpr int32_t ZSTD_E_FRAMEPARAM_UNSUPPORTED = 14;
constexpr int32_t ZSTD_E_WINDOW_TOO_LARGE = 16;
constexpr int32_t ZSTD_E_CORRUPTION = 20;
constexpr int32_t ZSTD_E_CHECKSUM_WRONG = 22;
constexpr int32_t ZSTD_E_DICT_WRONG = 32;

constexpr uint32_t RESUME_WORDS = 6;

// The streaming zstd decoder's checkpoint in device memory (written by zstd_kernel, sized by api.hip): a header of ZRES_HDR words, then the
// first ZSAVE_WORDS words of the kernel's LDS state (its decode tables; zstd.hip asserts that this is where `weights` starts), then slack.
constexpr uint32_t ZRES_HDR = 24;
constexpr uint32_t ZSAVE_WORDS = 2312;
constexpr size_t ZRES_BYTES = (size_t)(ZRES_HDR + ZSAVE_WORDS) * 4 + 64;

struct BatchArgs {
    const uint8_t *in_base;
    const uint64_t *in_off;
    const uint32_t *in_len;
    uint8_t *out_base;
    const uint64_t *out_off;
    const uint32_t *out_cap;
    uint32_t *out_len;
    uint32_t *in_used;
    int32_t *status;
    uint32_t n;
    int32_t format;
    unsigned long long *stats;  // diagnostic builds only (-DCHIP_STATS): 16 words per unit, else nullptr
    // inflate, streaming decoder only (else nullptr): RESUME_WORDS words per unit, in and out --
    //  [0] bit offset (in the unit's input as it is handed over) of the last block boundary reached, 0 = start from the beginning
    //  [1] output bytes produced up to there, as an offset into the unit's output range
    //  [2] wrapper kind (0 raw, 1 zlib, 2 gzip)
    //  [3] running check value (Adler-32 / CRC-32) over the first [4] bytes of the stream's output
    //  [4] output bytes the running check covers, counted from the start of the stream
    //  [5] output bytes the caller has dropped in front of the output range (offset 0 = stream byte [5]), low 32 bits of
    //      the stream's output count = [5] + offset
    // A later call over the unconsumed input (the caller may drop input in front of the boundary and output in front of the
    // 32 KiB window, adjusting [0], [1] and [5]) continues from that boundary.
    uint32_t *resume;
    // Routed batches (CHIP_FMT_DETECT): the kernel works on units sel[0 .. *sel_n) instead of 0 .. n (both device
    // pointers, written by route_kernel earlier on the same stream); nullptr = all n units in index order.
    const uint32_t *sel = nullptr;
    const uint32_t *sel_n = nullptr;
    uint32_t flags = 0;  // CHIP_F_* of chip_decode_batch_ex (include/compu_hip.h)
};
constexpr uint32_t F_COMPU_STATUS = 1u;  // = CHIP_F_COMPU_STATUS

// ---- the two-kernel inflate pipeline (inflate.hip: tokens_kernel, lz77.hip: lz77_kernel) -----------------------------------------
// tokens_kernel decodes a unit's bit stream into 32-bit tokens in stream order (token arena in HBM) and leaves a record per unit;
// lz77_kernel executes a unit's tokens in an LDS image of the unit's output (one workgroup per unit) and stores the image.  A unit
// that does not end cleanly, or does not fit the image, goes to the one-kernel path (inflate_kernel) through a fallback list.
// token: [8:0] literal byte, or match length 3..258; [9] match; [25:10] match distance - 1
constexpr uint32_t PIPE_MAXSEG = 24;                     // segments of a unit's record
constexpr uint32_t PIPE_REC_WORDS = 8 + 2 * PIPE_MAXSEG;  // words per unit record
constexpr uint32_t PIPE_ARENA_WORDS = 16384;             // tokens_kernel takes arena space in pieces of this many words (>= 64 lanes x 256 row tokens)
constexpr uint32_t PIPE_IMAGE_BYTES = 65536;             // lz77_kernel's image: units with more output take the one-kernel path
// unit record: [0] state (PIPE_ST_*), [1] number of segments, [2] wrapper kind (0 raw, 1 zlib, 2 gzip), [3] the trailer's check value,
// [4] gzip ISIZE, [5] input bytes used, [6..7] reserved; then per segment two words: [0] tokens: first word in the arena, stored
// bytes: offset in the unit's input; [1] count (tokens / bytes) | PIPE_SEG_STORED
constexpr uint32_t PIPE_ST_FALLBACK = 0, PIPE_ST_TOKENS = 1;
constexpr uint32_t PIPE_SEG_STORED = 1u << 31;
struct PipeScratch {
    uint32_t *rec;         // PIPE_REC_WORDS per unit (indexed by unit)
    uint32_t *arena;       // token arena
    uint32_t arena_words;  // its size
    uint32_t *counters;    // [0] tokens_kernel's unit counter, [1] arena words taken, [2] fallback count, [3] inflate_kernel's unit counter
    uint32_t *fallback;    // unit indices for inflate_kernel (n entries)
};
hipError_t launch_lz77(const BatchArgs &a, const PipeScratch &p, hipStream_t stream);

// launchers (each only enqueues on `stream`)
hipError_t launch_inflate(const BatchArgs &a, hipStream_t stream);
hipError_t release_inflate_scratch();  // frees the cached token scratch of the current device (after a device sync)
void release_inflate_scratch_of(hipStream_t stream);
hipError_t release_deflate_scratch();  // the encoder's token scratch (dynamic levels), same rules
void release_deflate_scratch_of(hipStream_t stream);  // the same for one (drained) stream of the current device
hipError_t launch_zstd_decode(const BatchArgs &a, int window_log_max, hipStream_t stream);
// Detection-driven router of a mixed batch: appends the index of every gzip / zlib unit to sel_inflate and of every zstd
// frame to sel_zstd (counts[0], counts[1], zeroed by the call) and answers units that are neither at once.
hipError_t launch_route(const BatchArgs &a, uint32_t *sel_inflate, uint32_t *sel_zstd, uint32_t *counts, hipStream_t stream);
// device scratch of a routed batch on `stream`, cached with the inflate slot: two index lists of n entries + 2 counters
hipError_t route_scratch(hipStream_t stream, size_t n, uint32_t **sel_inflate, uint32_t **sel_zstd, uint32_t **counts);
// a whole CHIP_FMT_DETECT batch (router + both decoders over their lists) under one lock of the (device, stream) slot
hipError_t launch_routed(const BatchArgs &a, hipStream_t stream);
hipError_t launch_detect(size_t n, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len, int32_t *kind,
                         hipStream_t stream);
hipError_t launch_deflate_l1(const BatchArgs &a, int level, uint32_t flags, uint32_t check_seed, uint64_t total_before,
                             uint32_t *check_out, hipStream_t stream);

// Detection::detect (src/decoder/mod.rs:28-114) on device: first 2-4 bytes of a unit -> CHIP_DETECT_*.
// The FLG table of mod.rs:44-55 is packed one word per CINFO; the 0x68 row never matches in the
// reference (mod.rs:80-82 lacks the `return`) and is kept that way.
__device__ __forceinline__ int32_t detect_kind(const uint8_t *b, uint32_t len)
{
    if (len < 2) return CHIP_DETECT_NONE;
    const uint32_t b0 = b[0], b1 = b[1];
    if (b0 == 0x1f && b1 == 0x8b) return CHIP_DETECT_GZIP;
    if (((b0 << 8) | b1) % 31 == 0 && (b0 & 0x8f) == 0x08 && b0 != 0x68) {
        const uint32_t rows[8] = {0x1d5b99d7u, 0x195795d3u, 0x155391cfu, 0x114f8dcbu, 0x0d4b89c7u, 0x094785c3u, 0x054381deu, 0x015e9cdau};
        const uint32_t r = rows[b0 >> 4];
        if (b1 == (r >> 24) || b1 == ((r >> 16) & 0xff) || b1 == ((r >> 8) & 0xff) || b1 == (r & 0xff)) return CHIP_DETECT_ZLIB;
    }
    if (len < 4) return CHIP_DETECT_NONE;
    return (b0 == 0x28 && b1 == 0xb5 && b[2] == 0x2f && b[3] == 0xfd) ? CHIP_DETECT_ZSTD : CHIP_DETECT_UNKNOWN;
}

// address spaces: LDS pointers keep theirs through calls and selects (a generic pointer to LDS becomes flat accesses), pointers
// into HBM keep theirs through scalar round trips
#define LDS_AS __attribute__((address_space(3)))
#define GAS __attribute__((address_space(1)))

// ---- wavefront helpers (wave = 64 lanes, one wave per workgroup in the codec kernels) ----------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ uint32_t rdfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// The same for a pointer: tells the compiler that a value every lane holds alike is wave-uniform, so that it lives in
// scalar registers and loops over it become scalar branches instead of exec-mask loops.
template <typename T>
__device__ __forceinline__ T *rdfirst_ptr(T *p)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    return (T *)(uintptr_t)(((uint64_t)rdfirst((uint32_t)(v >> 32)) << 32) | rdfirst((uint32_t)v));
}

// DPP lane moves within the wave (gfx9 controls): lanes without a source keep 0
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// inclusive prefix sum across the wave: Hillis-Steele inside each row of 16 lanes (row_shr:1,2,4,8),
// then the row totals are carried over with row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += dpp_or_zero<0x111>(v);
    v += dpp_or_zero<0x112>(v);
    v += dpp_or_zero<0x114>(v);
    v += dpp_or_zero<0x118>(v);
    v += dpp_or_zero<0x142, 0xa>(v);
    v += dpp_or_zero<0x143, 0xc>(v);
    return v;
}

// inclusive running maximum across the wave (same DPP ladder; 0 is the identity for unsigned max)
__device__ __forceinline__ uint32_t wave_incl_max_scan(uint32_t v)
{
    uint32_t t;
    t = dpp_or_zero<0x111>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x112>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x114>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x118>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x142, 0xa>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x143, 0xc>(v); v = v > t ? v : t;
    return v;
}

// value of the previous lane (lane 0 gets 0): DPP wave_shr:1
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return dpp_or_zero<0x138>(v); }

// LDS written by some lanes of the (single) wave, read by others: order + visibility
#define WSYNC() __syncthreads()
// The same for LDS only: leaves global loads and stores in flight (a one-wave workgroup needs no s_barrier, LDS
// operations of a wave complete in order; the memory clobber keeps the compiler from moving or forwarding LDS accesses)
#define LSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

}  // namespace chip
