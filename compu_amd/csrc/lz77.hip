// LZ77 execution for gfx950 (MI355X): one workgroup of eight wavefronts executes the token stream of one unit.
//
// Second half of the two-kernel inflate pipeline (chip_internal.h): tokens_kernel (inflate.hip) has turned a unit's bit stream into
// 32-bit tokens in stream order; this kernel turns the tokens into bytes -- the LZ77 copy of RFC 1951 sec. 3.2.3 and the trailer
// checks of RFC 1950 / 1952, i.e. the rest of what compu reaches through sys::inflate (src/decoder/mod.rs:470).
//
// Data flow per workgroup: the unit's whole output (at most 64 KiB) is assembled in LDS -- every LZ77 source is an LDS read, nothing
// of a unit's output is read back from HBM -- and leaves with 16-byte stores once the stream has been executed and checked.
//   * The token stream is cut into blocks of 256 tokens; wave w takes blocks w, w + 8, ...  A block's first output offset comes
//     from the block before it through LDS: each wave sums its block's output lengths with wave prefix sums and hands the end on at
//     once, before it executes the block.  Literals are dropped into the image by their lanes; matches go to the wave's pool.
//   * Beside the image lies a bit map, one bit per byte: the byte is final.  A literal's lane sets its bit; a match is copied when
//     the bits of its source are all set, and then sets the bits of its destination.  So the eight waves never wait for each
//     other's blocks: a match whose source is still being produced (the eight blocks in flight span about 6 KB of output: with
//     distances spread over 32 KiB that is one match in ten) stays in the pool and is looked at again with the next ones, whatever
//     the order in which the sources appear.  The pool is swept 64 entries at a time, one per lane.
// A unit that does not fit (more than 64 KiB of output, a distance reaching in front of the output, a failed trailer check, more
// output than the caller's capacity) is left untouched and handed to inflate_kernel through the fallback list: that kernel
// is the one that knows every status and position compu reports (src/decoder/mod.rs:475-483).
#include "chip_internal.h"
#include "wave_checksums.h"

namespace chip {

namespace {

// Diagnostic build (-DCHIP_STATS): per-unit event counters of lz77_kernel, summed over the waves into BatchArgs::stats (24 words per
// unit): 0 pool passes, 1 passes that copied nothing, 2 matches copied by their lanes, 3 matches copied by the wave, 4 trips that
// found the block's place not known yet, 5 trips blocked with an empty pool, 6 cycles in pool passes, 7 cycles in token groups,
// 8 cycles in block set-up (scans, waiting for the place), 9 cycles asleep, 10 cycles total (per wave), 11 entries looked at
#ifdef CHIP_STATS
#define XSTAT_ADD(i, v) (xst[i] += (unsigned long long)(v))
#define XSTAT_T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); xst[i] += n_ - xt0; xt0 = n_; } while (0)
#else
#define XSTAT_ADD(i, v)
#define XSTAT_T(i)
#endif

constexpr uint32_t X_NW = 8;          // waves per workgroup
constexpr uint32_t X_THREADS = 64 * X_NW;
constexpr uint32_t X_BLK = 256;       // tokens per block
constexpr uint32_t X_SBLK = 4096;     // stored bytes per block
constexpr uint32_t X_MAXBLK = 336;    // blocks per unit: 65536 tokens / 256, 65536 stored bytes / 4096, one partial block per segment
constexpr uint32_t X_POOL = 128;      // waiting matches per wave
#ifndef CHIP_X_PASS_AT
#define CHIP_X_PASS_AT 40
#endif
constexpr uint32_t X_PASS_AT = CHIP_X_PASS_AT;  // a pass over the pool when it holds this many (<= X_POOL - 64: a group may add 64)
constexpr uint32_t X_NOTYET = 0xffffffffu;
constexpr uint32_t X_COPY_MAX = 31;   // bytes a lane copies for its match (16 + 8 + 4 + 2 + 1); longer (and self-overlapping) matches are copied by the whole wave
constexpr uint32_t X_BMW = PIPE_IMAGE_BYTES / 32;

struct __attribute__((packed)) U32u { uint32_t v; };
struct __attribute__((packed)) U16u { uint16_t v; };
typedef LDS_AS uint8_t lds_u8;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));

struct alignas(16) XLds {
    uint8_t img[PIPE_IMAGE_BYTES + 32];   // the unit's output by offset (+ room for 16-byte reads that run past a source's end)
    uint32_t bm[X_BMW + 12];              // bit b of word w: image byte 32 w + b is final; at the end: CRC tables
    uint32_t pa[X_NW][X_POOL];            // per wave, waiting match: first output offset | distance << 16 ...
    uint16_t pl[X_NW][X_POOL];            // ... and its length
    uint32_t blk_len[X_MAXBLK + 8];       // [8 + b] output bytes of block b (X_NOTYET until its wave has added them up); [0..7] = 0
    uint32_t total;                       // the unit's output bytes (set by the last block's wave)
    uint32_t seg[2 * PIPE_MAXSEG];
    uint32_t part[2 * X_NW];
    uint32_t bad;
    uint32_t trash[64 + 4];               // lane l's word for the stores it must not make (see the kernel)
};
static_assert(sizeof(XLds) <= 80 * 1024, "two workgroups per CU");
static_assert(sizeof(((XLds *)0)->bm) >= 2048 * 4, "the CRC tables take the bit map's place");

// LDS words that other waves write / read while this wave runs: the access keeps its address space (through a generic pointer it is
// a flat load with system scope) and is not cached in registers or moved by the compiler
__device__ __forceinline__ uint32_t lds_peek(const uint32_t *p)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(const LDS_AS uint32_t *)p) : "memory");
    return v;
}
// two consecutive words
__device__ __forceinline__ void lds_peek2(const uint32_t *p, uint32_t &w0, uint32_t &w1)
{
    u32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(const LDS_AS uint32_t *)p) : "memory");
    w0 = v.x;
    w1 = v.y;
}
__device__ __forceinline__ void lds_poke(uint32_t *p, uint32_t v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)(LDS_AS uint32_t *)p), "v"(v) : "memory");
}

// exactly n (1..16) bytes of v to the LDS address d
__device__ __forceinline__ void lds_put(lds_u8 *d, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t n)
{
    if (n >= 4) ((LDS_AS U32u *)d)->v = v0;
    if (n >= 8) ((LDS_AS U32u *)(d + 4))->v = v1;
    if (n >= 12) ((LDS_AS U32u *)(d + 8))->v = v2;
    if (n >= 16) ((LDS_AS U32u *)(d + 12))->v = v3;
    uint32_t w = n < 4 ? v0 : n < 8 ? v1 : n < 12 ? v2 : v3;
    lds_u8 *t = d + (n & 12u);
    if (n < 16 && (n & 2u)) {
        ((LDS_AS U16u *)t)->v = (uint16_t)w;
        w >>= 16;
        t += 2;
    }
    if (n < 16 && (n & 1u)) *t = (uint8_t)w;
}

// (off % d) for off, d < 512 without an integer divide
__device__ __forceinline__ uint32_t small_mod(uint32_t off, uint32_t d)
{
    const uint32_t q = (uint32_t)(((float)off + 0.5f) * __builtin_amdgcn_rcpf((float)d));
    return off - q * d;
}

// bits [a, b) of the image's bit map as seen by the word that starts at bit 32 w (a < b)
__device__ __forceinline__ uint32_t word_mask(uint32_t a, uint32_t b, uint32_t w)
{
    const uint32_t lo = 32u * w, hi = lo + 32u;
    const uint32_t s = a > lo ? a - lo : 0u;  // first bit inside the word
    if (a >= hi || b <= lo) return 0u;
    const uint32_t m = 0xffffffffu << s;
    return b >= hi ? m : m & ((1u << (b - lo)) - 1u);
}

// ---- trailer checks over the image, by the whole workgroup -----------------------------------------------------------------------
// thread t gets the right-aligned chunk [beg, end) of [0, n); chunk = power of two >= n / X_THREADS
__device__ __forceinline__ void wg_chunk(uint32_t n, uint32_t t, uint32_t &chunk_log2, uint32_t &beg, uint32_t &end)
{
    const uint32_t per = (n + X_THREADS - 1u) / X_THREADS;
    chunk_log2 = per <= 1 ? 0u : 32u - (uint32_t)__clz((int)(per - 1));
    const int64_t chunk = 1ll << chunk_log2;
    const int64_t e = (int64_t)n - (int64_t)(X_THREADS - 1u - t) * chunk, b = e - chunk;
    end = e > 0 ? (uint32_t)e : 0u;
    beg = b > 0 ? (uint32_t)b : 0u;
}

// CRC-32 (RFC 1952 sec. 8) of img[0..n); the bit map's place holds the tables (2048 words).  Uniform result.
__device__ uint32_t wg_crc32(XLds &L, uint32_t n)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t *const tab = L.bm;
    __syncthreads();
    if (tid < 256) {
        uint32_t c = tid;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
        tab[tid] = c;
    }
    __syncthreads();
    for (uint32_t k = 1; k < 8; k++) {
        if (tid < 256) {
            const uint32_t c = tab[(k - 1) * 256 + tid];
            tab[k * 256 + tid] = (c >> 8) ^ tab[c & 0xffu];
        }
        __syncthreads();
    }
    uint32_t lg, beg, end;
    wg_chunk(n, tid, lg, beg, end);
    uint32_t c = (beg == 0 && end > 0) ? 0xffffffffu : 0u;  // the thread that owns byte 0 carries the start value
    if (n == 0 && tid == X_THREADS - 1u) c = 0xffffffffu;
    uint32_t k = beg;
    while (k < end && (k & 7u) != 0) {
        c = tab[(c ^ L.img[k]) & 0xffu] ^ (c >> 8);
        k++;
    }
    for (; k + 8 <= end; k += 8) {
        const uint2 d = *(const uint2 *)(L.img + k);
        const uint32_t lo = d.x ^ c, hi = d.y;
        c = tab[7 * 256 + (lo & 0xffu)] ^ tab[6 * 256 + ((lo >> 8) & 0xffu)] ^ tab[5 * 256 + ((lo >> 16) & 0xffu)] ^ tab[4 * 256 + (lo >> 24)] ^
            tab[3 * 256 + (hi & 0xffu)] ^ tab[2 * 256 + ((hi >> 8) & 0xffu)] ^ tab[1 * 256 + ((hi >> 16) & 0xffu)] ^ tab[hi >> 24];
    }
    for (; k < end; k++) c = tab[(c ^ L.img[k]) & 0xffu] ^ (c >> 8);
    // tree combine: state(A||B) = state(A) * x^(8|B|) + raw(B); at level j the right block has 2^j chunks
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const uint32_t left = (uint32_t)__shfl_up((int)c, 1 << j, 64);
        const uint32_t comb = multmodp(X2N[(3 + lg + j) & 31], left) ^ c;
        if ((lane & ((2u << j) - 1)) == ((2u << j) - 1)) c = comb;
    }
    if (lane == 63) L.part[wave] = c;
    __syncthreads();
    c = lane < X_NW ? L.part[lane] : 0u;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t left = (uint32_t)__shfl_up((int)c, 1 << j, 64);
        const uint32_t comb = multmodp(X2N[(3 + lg + 6 + j) & 31], left) ^ c;
        if ((lane & ((2u << j) - 1)) == ((2u << j) - 1)) c = comb;
    }
    __syncthreads();
    return ~rdlane(c, X_NW - 1);
}

// Adler-32 (RFC 1950 sec. 9) of img[0..n).  Uniform result.
__device__ uint32_t wg_adler32(XLds &L, uint32_t n)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t lg, beg, end;
    wg_chunk(n, tid, lg, beg, end);
    uint32_t a = 0, b = 0;
    for (uint32_t k = beg; k < end; k++) {  // a chunk is at most 128 bytes: b < 128 * 255 * 128
        a += L.img[k];
        b += a;
    }
    // B = n + sum_i (b_i + a_i * bytes behind chunk i), A = 1 + sum_i a_i
    const uint32_t after = n - end;
    const uint32_t term = (uint32_t)(((uint64_t)b + (uint64_t)a * after) % 65521u);
    const uint32_t sa = wave_incl_scan(a), sb = wave_incl_scan(term);  // < 64 * 128 * 255 and < 64 * 65521: no overflow
    __syncthreads();
    if (lane == 63) {
        L.part[2 * wave] = sa;
        L.part[2 * wave + 1] = sb;
    }
    __syncthreads();
    uint32_t A = 1, B = n % 65521u;
    for (uint32_t w = 0; w < X_NW; w++) {
        A += L.part[2 * w];
        B = (B + L.part[2 * w + 1] % 65521u) % 65521u;
    }
    __syncthreads();
    return (B << 16) | (A % 65521u);
}

}  // namespace

// (no waves-per-SIMD bound: with one the compiler splits the register file in two halves and spills into a0.., the token sets' registers; as it
// is the kernel takes about 80 vector registers + the 12 accumulation registers: four waves per SIMD)
__global__ __launch_bounds__(X_THREADS) void lz77_kernel(BatchArgs a, PipeScratch p)
{
    __shared__ XLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = rdfirst(tid >> 6);
    const uint32_t limit = a.sel_n ? *a.sel_n : a.n;
    if (blockIdx.x >= limit) return;
    const uint32_t u = a.sel ? a.sel[blockIdx.x] : blockIdx.x;
    const uint32_t *const rec = p.rec + (size_t)u * PIPE_REC_WORDS;
    if (rec[0] != PIPE_ST_TOKENS) return;  // the unit is on the fallback list already
    const uint32_t nseg = rdfirst(rec[1]);
    for (uint32_t k = tid; k < X_MAXBLK + 8; k += X_THREADS) L.blk_len[k] = k < 8 ? 0u : X_NOTYET;
    if (tid == 0) L.total = nseg ? X_NOTYET : 0u;
    for (uint32_t k = tid; k < X_BMW + 12; k += X_THREADS) L.bm[k] = 0;
    if (tid < 2 * nseg) L.seg[tid] = rec[8 + tid];
    if (tid == 0) L.bad = 0;
    __syncthreads();
    uint32_t nblk = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        const uint32_t cw = rdfirst(L.seg[2 * s + 1]), cnt = cw & ~PIPE_SEG_STORED;
        nblk += (cw & PIPE_SEG_STORED) ? (cnt + X_SBLK - 1) / X_SBLK : (cnt + X_BLK - 1) / X_BLK;
    }
    const uint8_t *const gin = a.in_base + a.in_off[u];
    const bool fits = nblk <= X_MAXBLK;
    if (fits) {
        lds_u8 *const img = (lds_u8 *)L.img;
        const uint32_t img_a = (uint32_t)(uintptr_t)img;  // LDS byte addresses: a store that a lane must not make goes to the lane's trash word
        const uint32_t bm_a = (uint32_t)(uintptr_t)(LDS_AS uint32_t *)L.bm;
        LDS_AS uint32_t *const pa = (LDS_AS uint32_t *)L.pa[wave];
        LDS_AS uint16_t *const pl = (LDS_AS uint16_t *)L.pl[wave];
        const uint32_t trash_a = (uint32_t)(uintptr_t)(LDS_AS uint32_t *)L.trash + 4u * lane;
        uint32_t np = 0;  // waiting matches of this wave
        // this wave's blocks: segment s, block j of it, number gb among the unit's blocks
        uint32_t it_s = 0, it_j = wave, it_gb0 = 0;
        struct Blk {
            uint32_t valid, stored;
            uint32_t src, n, gb;
        };
        auto next_block = [&]() -> Blk {
            Blk b{0, 0, 0, 0, 0};
            while (it_s < nseg) {
                const uint32_t off = rdfirst(L.seg[2 * it_s]), cw = rdfirst(L.seg[2 * it_s + 1]), cnt = cw & ~PIPE_SEG_STORED;
                const uint32_t st = cw >> 31;
                const uint32_t sh = st ? 12u : 8u, per = 1u << sh, nb = (cnt + per - 1u) >> sh;  // (X_SBLK = 2^12, X_BLK = 2^8)
                if (it_j < nb) {
                    b.valid = 1;
                    b.stored = st;
                    b.src = off + (it_j << sh);
                    b.n = cnt - (it_j << sh) < per ? cnt - (it_j << sh) : per;
                    b.gb = it_gb0 + it_j;
                    it_j += X_NW;
                    return b;
                }
                it_gb0 += nb;
                it_s++;
                it_j = (wave + X_NW - (it_gb0 % X_NW)) % X_NW;
            }
            return b;
        };
        // The tokens of a block (four per lane) are asked for three blocks ahead -- a load from HBM takes longer than a block -- into one of
        // three sets of four accumulation registers (a0..a11): the compiler never touches those, so nothing copies a register while its
        // load is in flight, and the loads, issued in assembly, are waited for by count (the compiler would wait for all of them at
        // the first use of any).  Always four loads per set, also for a block that has no tokens: the count stays the same.
#define X_ISSUE(R0, R1, R2, R3)                                                                                                                  \
    asm volatile("global_load_dword " R0 ", %0, off\n\tglobal_load_dword " R1 ", %1, off\n\tglobal_load_dword " R2 ", %2, off\n\tglobal_load_dword " R3 \
                 ", %3, off" ::"v"(a0),                                                                                                          \
                 "v"(a1), "v"(a2), "v"(a3)                                                                                                       \
                 : "memory", R0, R1, R2, R3)
#define X_TAKE(R0, R1, R2, R3) \
    asm volatile("v_accvgpr_read_b32 %0, " R0 "\n\tv_accvgpr_read_b32 %1, " R1 "\n\tv_accvgpr_read_b32 %2, " R2 "\n\tv_accvgpr_read_b32 %3, " R3 : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)::"memory")
        auto issue = [&](const Blk &b, uint32_t set) {
            const bool tok = b.valid && !b.stored;
            GAS const uint32_t *base = (GAS const uint32_t *)p.arena + (tok ? (size_t)b.src : (size_t)0);
            const uint32_t nn = tok ? b.n : 1u;  // (lanes behind the block's end read its first token again)
            GAS const uint32_t *a0 = base + (lane < nn ? lane : 0u), *a1 = base + (64u + lane < nn ? 64u + lane : 0u);
            GAS const uint32_t *a2 = base + (128u + lane < nn ? 128u + lane : 0u), *a3 = base + (192u + lane < nn ? 192u + lane : 0u);
            if (set == 0) X_ISSUE("a0", "a1", "a2", "a3");
            else if (set == 1) X_ISSUE("a4", "a5", "a6", "a7");
            else X_ISSUE("a8", "a9", "a10", "a11");
        };
        Blk cur = next_block();
        uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (cur.valid && !cur.stored) {
            if (lane < cur.n) t0 = ((GAS const uint32_t *)p.arena)[(size_t)cur.src + lane];
            if (64u + lane < cur.n) t1 = ((GAS const uint32_t *)p.arena)[(size_t)cur.src + 64u + lane];
            if (128u + lane < cur.n) t2 = ((GAS const uint32_t *)p.arena)[(size_t)cur.src + 128u + lane];
            if (192u + lane < cur.n) t3 = ((GAS const uint32_t *)p.arena)[(size_t)cur.src + 192u + lane];
        }
        Blk ba = next_block();
        issue(ba, 0);
        Blk bb = next_block();
        issue(bb, 1);
        Blk bc = next_block();
        issue(bc, 2);
        uint32_t ph = 0;  // the set that holds the next block's tokens
#ifdef CHIP_STATS
        unsigned long long xst[12] = {0}, xt0 = __builtin_readcyclecounter();
        const unsigned long long xstart = xt0;
#endif
        uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;  // the groups' first offsets inside the block
        uint32_t run = 0, B0 = 0, g = 0, passno = 0;
        uint32_t placed = 0, scanned = 0, lenpub = 0, lastB0 = 0;
        // One loop, one step per trip: the wave either takes the next step of its current block (place the block, then its four groups
        // of 64 tokens) or, when it cannot (the block's first offset is not known yet, the pool is full, nothing is left but the pool),
        // makes a pass over the pool.  (One copy of each piece of code: the kernel stays small; stores that a lane must not make go to its
        // trash word instead of sitting in a branch: exec-mask bookkeeping is scalar work, and the scalar unit is shared by the CU.)
        for (;;) {
            uint32_t blocked = 0;
            if (!cur.valid) {
                if (!np) break;
                blocked = 1;
            } else if (!placed) {
                if (!scanned && !cur.stored) {
                    // output lengths of the block's tokens, places inside the block
                    uint32_t o, incl;
                    o = (t0 & 512u) ? t0 & 0x1ffu : 1u; o = lane < cur.n ? o : 0u; incl = wave_incl_scan(o); s0 = incl - o; run = rdlane(incl, 63u);
                    o = (t1 & 512u) ? t1 & 0x1ffu : 1u; o = 64u + lane < cur.n ? o : 0u; incl = wave_incl_scan(o); s1 = run + incl - o; run += rdlane(incl, 63u);
                    o = (t2 & 512u) ? t2 & 0x1ffu : 1u; o = 128u + lane < cur.n ? o : 0u; incl = wave_incl_scan(o); s2 = run + incl - o; run += rdlane(incl, 63u);
                    o = (t3 & 512u) ? t3 & 0x1ffu : 1u; o = 192u + lane < cur.n ? o : 0u; incl = wave_incl_scan(o); s3 = run + incl - o; run += rdlane(incl, 63u);
                }
                if (cur.stored) run = cur.n;
                scanned = 1;
                if (!lenpub) {
                    // the block's length goes out at once: the waves behind place themselves by adding up the lengths in front of
                    // them, none waits for another's place
                    if (lane == 0) {
                        lds_poke(&L.blk_len[8u + cur.gb], run);
                    }
                    lenpub = 1;
                }
                // first offset = this wave's last block's first offset + the lengths of the eight blocks since (its own and the other
                // waves' seven; the array starts with eight zeros for the unit's first blocks)
                uint32_t l0, l1, l2, l3, l4, l5, l6, l7;
                u32x2 l01, l23, l45, l67;
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:2 offset1:3\n\tds_read2_b32 %2, %4 offset0:4 offset1:5\n\tds_read2_b32 %3, %4 offset0:6 offset1:7\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(l01), "=&v"(l23), "=&v"(l45), "=&v"(l67)
                             : "v"((uint32_t)(uintptr_t)(LDS_AS uint32_t *)&L.blk_len[cur.gb])
                             : "memory");
                l0 = rdfirst(l01.x), l1 = rdfirst(l01.y), l2 = rdfirst(l23.x), l3 = rdfirst(l23.y), l4 = rdfirst(l45.x), l5 = rdfirst(l45.y), l6 = rdfirst(l67.x), l7 = rdfirst(l67.y);
                if ((l0 | l1 | l2 | l3 | l4 | l5 | l6 | l7) >> 31) {  // (a length is below 2^31; X_NOTYET is all ones)
                    blocked = 1;
                    XSTAT_ADD(4, 1);
                } else {
                    B0 = lastB0 + l0 + l1 + l2 + l3 + l4 + l5 + l6 + l7;
                    lastB0 = B0;
                    if (cur.gb + 1u == nblk && lane == 0) lds_poke(&L.total, B0 + run);
                    if (B0 + run > PIPE_IMAGE_BYTES) {
                        if (lane == 0) atomicOr(&L.bad, 1u);
                        break;
                    }
                    placed = 1;
                    g = 0;
                    if (cur.stored) {
                        const uint8_t *sp = gin + cur.src;
                        for (uint32_t k = 16u * lane; k < cur.n; k += 1024u) {
                            if (k + 16u <= cur.n) {
                                const u32x4_u v = *(GAS const u32x4_u *)(sp + k);
                                lds_put(img + B0 + k, v.x, v.y, v.z, v.w, 16u);
                            } else {
                                for (uint32_t q = k; q < cur.n; q++) img[B0 + q] = sp[q];
                            }
                        }
                        LSYNC();
                        for (uint32_t w = lane; w < X_SBLK / 32 + 1; w += 64u) {  // the run's bits
                            const uint32_t mk = word_mask(B0, B0 + cur.n, (B0 >> 5) + w);
                            if (mk) atomicOr(&L.bm[(B0 >> 5) + w], mk);
                        }
                        g = 4;  // (nothing else to do for this block)
                    }
                }
            }
            // A pass as soon as the pool holds most of a wave's worth: what waits in a pool is not final, and other waves' matches wait for it
            // (with eight waves each holding a full pool most of the last 8 KB of output would be waiting).
            if (!blocked && cur.valid && g < 4u && 64u * g < cur.n && np >= X_PASS_AT) blocked = 1;
            XSTAT_T(8);
            if (blocked) {
                // ---- one pass over the pool's oldest 64 entries, one per lane: the matches whose source bytes are final are copied, the
                // others go to the back of the queue.  A lane copies up to 15 bytes; every fourth pass up to 31.
                uint32_t progress = 0;
                if (np) {
                    const uint32_t n = np < 64u ? np : 64u, rest = np - n;
                    const bool wide = (passno & 3u) == 0;
                    passno++;
                    uint32_t ea = 0, len = 0;
                    if (lane < n) {
                        ea = pa[lane];
                        len = pl[lane];
                    }
                    uint32_t ta = 0, tl = 0;
                    if (lane < rest) {
                        ta = pa[64u + lane];
                        tl = pl[64u + lane];
                    }
                    const uint32_t x = ea & 0xffffu, dist = ea >> 16;
                    const uint32_t src = x - dist;
                    // the bits of [src, src + len) (at most 32 of them, in two words)
                    uint32_t w0, w1;
                    lds_peek2(&L.bm[src >> 5], w0, w1);
                    const uint32_t v = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (src & 31u));
                    const uint32_t needm = (1u << (len & 31u)) - 1u;
                    const bool rdy = len != 0 && len <= (wide ? X_COPY_MAX : 15u) && dist >= len && (~v & needm) == 0u;
                    if (rdy) {
                        // exactly len bytes: 16 (wide passes), 8, 4, 2, 1 as the bits of len say; a store that is not due goes to the lane's
                        // trash word (a choice of address instead of a branch: exec-mask bookkeeping is scalar work, and there is a lot of it)
                        uint32_t sa = img_a + src, da = img_a + x;
                        if (wide) {
                            uint32_t v0, v1, v2, v3;
                            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                                         : "v"(sa)
                                         : "memory");
                            const bool big = (len & 16u) != 0;
                            const uint32_t d0 = big ? da : trash_a;
                            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:8\n\tds_write_b32 %0, %4 offset:12" ::"v"(d0), "v"(v0), "v"(v1),
                                         "v"(v2), "v"(v3)
                                         : "memory");
                            sa += len & 16u;
                            da += len & 16u;
                        }
                        uint32_t v0, v1, v2, v3;
                        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                                     : "v"(sa)
                                     : "memory");
                        const uint32_t d8 = (len & 8u) ? da : trash_a;
                        const uint32_t d4 = (len & 4u) ? da + (len & 8u) : trash_a;
                        const uint32_t w4 = (len & 8u) ? v2 : v0;
                        const uint32_t ww = (len & 8u) ? ((len & 4u) ? v3 : v2) : ((len & 4u) ? v1 : v0);  // the dword that holds bytes (len & 12) ..
                        const uint32_t d2 = (len & 2u) ? da + (len & 12u) : trash_a;
                        const uint32_t d1 = (len & 1u) ? da + (len & 14u) : trash_a;
                        const uint32_t w1b = (len & 2u) ? ww >> 16 : ww;
                        asm volatile("ds_write_b32 %0, %4\n\tds_write_b32 %0, %5 offset:4\n\tds_write_b32 %1, %6\n\tds_write_b16 %2, %7\n\tds_write_b8 %3, %8\n\ts_waitcnt lgkmcnt(0)" ::"v"(d8), "v"(d4),
                                     "v"(d2), "v"(d1), "v"(v0), "v"(v1), "v"(w4), "v"(ww), "v"(w1b)
                                     : "memory");
                        // (the bytes are in the image before their bits say so)
                        const uint64_t m = (uint64_t)needm << (x & 31u);
                        const uint32_t wa = bm_a + 4u * (x >> 5);
                        const uint32_t a1 = (uint32_t)(m >> 32) ? wa + 4u : trash_a;
                        asm volatile("ds_or_b32 %0, %2\n\tds_or_b32 %1, %3" ::"v"(wa), "v"(a1), "v"((uint32_t)m), "v"((uint32_t)(m >> 32)) : "memory");
                    }
                    bool done = rdy;
                    // long (more than 32 bytes) and self-overlapping matches, one at a time by the whole wave (lanes 0..9 look at / set the
                    // words of a range of up to 258 bits)
                    uint64_t cm = __ballot(len > X_COPY_MAX || dist < len);
                    while (cm) {
                        const uint32_t c = (uint32_t)__ffsll((long long)cm) - 1u;
                        cm &= cm - 1ull;
                        const uint32_t cx = rdlane(x, c), clen = rdlane(len, c), cdist = rdlane(dist, c);
                        const uint32_t csrc = cx - cdist, cn = clen < cdist ? clen : cdist;  // a self-overlapping match repeats its first `dist` bytes
                        const uint32_t fw = csrc >> 5;
                        const uint32_t mk = word_mask(csrc, csrc + cn, fw + lane);
                        const uint32_t have = lane < 10u ? lds_peek(&L.bm[fw + lane]) : 0u;
                        if (__any(lane < 10u && (have & mk) != mk)) continue;
                        for (uint32_t i = lane; i < clen; i += 64u) img[cx + i] = img[csrc + (cdist >= clen ? i : small_mod(i, cdist))];
                        LSYNC();
                        const uint32_t dw = cx >> 5;
                        const uint32_t dm = word_mask(cx, cx + clen, dw + lane);
                        if (lane < 10u && dm) atomicOr(&L.bm[dw + lane], dm);
                        if (lane == c) done = true;
                        XSTAT_ADD(3, 1);
                    }
                    // the queue moves up: what lay behind the 64 comes first, the entries that stay go behind it
                    const bool keep = len != 0 && !done;
                    const uint64_t km = __ballot(keep);
                    const uint32_t r = (uint32_t)__popcll(km);
                    if (lane < rest) {
                        pa[lane] = ta;
                        pl[lane] = (uint16_t)tl;
                    }
                    if (keep) {
                        const uint32_t q = rest + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                        pa[q] = ea;
                        pl[q] = (uint16_t)len;
                    }
                    LSYNC();
                    np = rest + r;
                    progress = r < n;
                    XSTAT_ADD(0, 1);
                    XSTAT_ADD(1, progress ? 0 : 1);
                    XSTAT_ADD(2, __popcll(__ballot(rdy)));
                    XSTAT_ADD(11, n);
                }
                XSTAT_T(6);
                if (!progress) {
                    // (the sources, or the block's place, are other waves' work)  A unit that cannot be finished here (L.bad: it takes the
                    // one-kernel path) is given up by every wave: bytes that are never produced would be waited for for ever.
                    if (rdfirst(lds_peek(&L.bad)) != 0u) break;
                    __builtin_amdgcn_s_sleep(2);
                    XSTAT_T(9);
                }
                continue;
            }
            if (g < 4u && 64u * g < cur.n) {
                // ---- group g of the block: literals into the image, matches into the pool
                const uint32_t tg = g == 0 ? t0 : g == 1 ? t1 : g == 2 ? t2 : t3;
                const uint32_t sg = g == 0 ? s0 : g == 1 ? s1 : g == 2 ? s2 : s3;
                const bool valid = 64u * g + lane < cur.n;
                const bool ismatch = valid && (tg & 512u) != 0;
                const uint32_t x = B0 + sg;
                const uint32_t len = tg & 0x1ffu, dist = __builtin_amdgcn_ubfe(tg, 10, 16) + 1u;
                const bool far = ismatch && dist > x;  // "invalid distance too far back": the one-kernel path reports it
                if (__any(far)) {
                    if (lane == 0) atomicOr(&L.bad, 2u);
                }
                const uint64_t mm = __ballot(ismatch);
                if (ismatch) {
                    const uint32_t q = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, np));
                    pa[q] = x | (dist << 16);
                    pl[q] = (uint16_t)len;
                }
                if (valid && !ismatch) {
                    // (the literal is in the image before its bit says so)
                    asm volatile("ds_write_b8 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tds_or_b32 %1, %3" ::"v"(img_a + x), "v"(bm_a + 4u * (x >> 5)), "v"(tg), "v"(1u << (x & 31u)) : "memory");
                }
                np += (uint32_t)__popcll(mm);
                g++;
                XSTAT_T(7);
                continue;
            }
            // ---- the block is done: on to the wave's next one; its tokens were asked for three blocks ago (two sets may still be on
            // their way), the set it leaves is filled again
            XSTAT_T(8);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            XSTAT_T(5);
            if (ph == 0) {
                cur = ba;
                X_TAKE("a0", "a1", "a2", "a3");
                ba = next_block();
                issue(ba, 0);
            } else if (ph == 1) {
                cur = bb;
                X_TAKE("a4", "a5", "a6", "a7");
                bb = next_block();
                issue(bb, 1);
            } else {
                cur = bc;
                X_TAKE("a8", "a9", "a10", "a11");
                bc = next_block();
                issue(bc, 2);
            }
            ph = ph == 2 ? 0 : ph + 1;
            placed = 0;
            scanned = 0;
            lenpub = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no token load is left in flight)
#ifdef CHIP_STATS
        xst[10] = __builtin_readcyclecounter() - xstart;
        if (a.stats && lane == 0)
            for (int k = 0; k < 12; k++) atomicAdd(&a.stats[(size_t)u * 24 + k], xst[k]);
#endif
    }
    __syncthreads();
    // ---- the whole stream is in the image: checks, then the stores
    const uint32_t total = fits ? L.total : 0u;
    const uint32_t cap = a.out_cap[u];
    bool ok = fits && L.bad == 0 && total != X_NOTYET && total <= PIPE_IMAGE_BYTES && total <= cap;
    const uint32_t wrap = rec[2];
    if (ok && wrap == 1) ok = wg_adler32(L, total) == rec[3];
    if (ok && wrap == 2) ok = wg_crc32(L, total) == rec[3] && rec[4] == total;
    if (!ok) {
        if (tid == 0) p.fallback[atomicAdd(&p.counters[2], 1u)] = u;
        return;
    }
    uint8_t *const gout = a.out_base + a.out_off[u];
    uint32_t head_b = (uint32_t)((16u - ((uintptr_t)gout & 15u)) & 15u);
    if (head_b > total) head_b = total;
    if (tid < head_b) gout[tid] = L.img[tid];
    const uint32_t body = (total - head_b) & ~15u;
    for (uint32_t k = 16u * tid; k < body; k += 16u * X_THREADS) {
        const lds_u8 *sp = (const lds_u8 *)L.img + head_b + k;
        u32x4 v;
        v.x = ((const LDS_AS U32u *)sp)->v;
        v.y = ((const LDS_AS U32u *)(sp + 4))->v;
        v.z = ((const LDS_AS U32u *)(sp + 8))->v;
        v.w = ((const LDS_AS U32u *)(sp + 12))->v;
        *(GAS u32x4 *)(gout + head_b + k) = v;
    }
    const uint32_t tail0 = head_b + body;
    if (tid < total - tail0) gout[tail0 + tid] = L.img[tail0 + tid];
    if (tid == 0) {
        a.out_len[u] = total;
        a.in_used[u] = rec[5];
        a.status[u] = CHIP_FINISHED;
    }
}

hipError_t launch_lz77(const BatchArgs &a, const PipeScratch &p, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz77_kernel, dim3(a.n), dim3(X_THREADS), 0, stream, a, p);
    return hipGetLastError();
}

}  // namespace chip
