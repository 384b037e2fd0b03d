// Batched DEFLATE encoder for gfx950 (MI355X): one wavefront encodes one unit.
//
// Replaces, per unit, what compu reaches through sys::deflate (src/encoder/mod.rs:352) for an encoder
// built by Interface::zlib_ng(ZlibOptions) (src/encoder/zlib_ng.rs:50-87): greedy hash matching in a
// 32 KiB window; level 1 writes one fixed-Huffman block (the shape of zlib-ng's deflate_quick; BASELINE
// configs[3]), levels 2..9 dynamic-Huffman blocks (4..9 with lazy choice); gzip / zlib wrappers with CRC-32 / Adler-32 trailers.
// compu's own tests pin the encoder by round trip and cross-API determinism only (tests/encoder.rs:10-78);
// the exact algorithm is stated in oracle/oracle_deflate.c and these kernels reproduce its output byte for byte.
//
// Per 64-position chunk: every lane hashes the 4 bytes at its position, looks the hash table (LDS,
// state before the chunk) up, measures the common prefix with the candidate (16-byte compares from
// HBM/L2), the table then takes the highest position per slot (16-bit entries), the greedy token choice is
// a scalar walk over the chosen matches.  Level 1: the tokens are packed at once with a wave prefix sum
// of their bit lengths into an LDS bit buffer that is flushed to HBM.  Levels 2..9: the tokens go to HBM
// scratch and are coded when the block closes (code lengths, header, second pass over the tokens).
#include "chip_internal.h"
#include "wave_checksums.h"

#include <map>
#include <mutex>

namespace chip {

namespace {

#ifndef CHIP_HASH_BITS
#define CHIP_HASH_BITS 12
#endif
constexpr int HASH_BITS = CHIP_HASH_BITS;
constexpr uint32_t MIN_MATCH = 4, MAX_MATCH = 258, MAX_DIST = 32768;
constexpr int OUT_DW = 192;  // LDS bit buffer, dwords (a chunk adds at most 62); with the table and lentab: 10 240 B = sixteen waves per CU

struct alignas(16) ELds {
    static constexpr int WAYS = 1;
    static constexpr uint32_t OBUF = OUT_DW + 64;
    uint16_t table[1 << HASH_BITS];  // low 16 bits of the highest position with that hash (oracle_deflate.c states the rule)
    uint32_t obuf[OBUF];
    uint32_t lentab[256];  // per match length 3..258: fixed-Huffman code with the extra bits | bit count << 16
};
static_assert(sizeof(ELds) <= 10240, "sixteen waves per CU");

// 4 input bytes at byte offset `off` of the dword-aligned view (little endian)
__device__ __forceinline__ uint32_t ld32(const uint32_t *g32, uint32_t total_dw, uint32_t off)
{
    uint32_t i = off >> 2;
    uint32_t d0 = i < total_dw ? g32[i] : 0u;
    uint32_t d1 = i + 1 < total_dw ? g32[i + 1] : 0u;
    return __builtin_amdgcn_alignbit(d1, d0, (off & 3u) * 8u);
}

// 16 bytes at any address (gfx950 does unaligned global loads)
struct __attribute__((packed, aligned(1))) U128u {
    uint32_t x, y, z, w;
};

// index of the first differing byte of two 16-byte strings, 16 when equal
__device__ __forceinline__ uint32_t first_diff16(const U128u a, const U128u b)
{
    const uint64_t lo = ((uint64_t)(a.y ^ b.y) << 32) | (a.x ^ b.x), hi = ((uint64_t)(a.w ^ b.w) << 32) | (a.z ^ b.z);
    const uint32_t klo = lo ? ((uint32_t)__ffsll((long long)lo) - 1u) >> 3 : 8u;
    const uint32_t khi = hi ? (((uint32_t)__ffsll((long long)hi) - 1u) >> 3) + 8u : 16u;
    return lo ? klo : khi;
}

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t n) { return __brev(v) >> (32 - n); }

// RFC 1951 sec. 3.2.5: length -> code 0..28 (symbol 257 + code), extra-bit count and value
__device__ __forceinline__ void len_parts(uint32_t len, uint32_t &lc, uint32_t &lext, uint32_t &lxv)
{
    // length code: 0..7 -> lengths 3..10; then groups of four per extra-bit count; 28 -> 258
    uint32_t lbase;
    if (len < 11) {
        lc = len - 3;
        lbase = len;
        lext = 0;
    } else if (len == 258) {
        lc = 28;
        lbase = 258;
        lext = 0;
    } else {
        uint32_t x = len - 3;                              // 8..254
        lext = (31u - (uint32_t)__clz((int)x)) - 2u;       // 1..5
        lc = 4u * lext + 4u + ((x >> lext) & 3u);
        lbase = 3u + ((4u + ((x >> lext) & 3u)) << lext);
    }
    lxv = len - lbase;
}

// distance -> code 0..29, extra-bit count and value
__device__ __forceinline__ void dist_parts(uint32_t dist, uint32_t &dc, uint32_t &dext, uint32_t &dxv)
{
    // codes come in pairs per extra-bit count: with h = the top bit of y = dist-1, the code is 2h + the bit below it
    // (y < 2: the code is y itself, which the same expression gives with the extra-bit count held at 0)
    const uint32_t y = dist - 1u;                               // 0..32767
    const uint32_t h = 31u - (uint32_t)__clz((int)(y | 1u));    // 0..14
    dext = (h > 1u ? h : 1u) - 1u;                              // 0..13
    dc = 2u * h + ((y >> dext) & 1u);
    dxv = y & ~(~0u << dext);
}

struct EncArgs {
    BatchArgs b;
    int32_t level;
    uint32_t flags;       // bit0 wrapper header, bit1 wrapper trailer, bit2 final block (else sync marker), [10:8] CHIP_STRATEGY_*
    uint32_t check_seed;  // running CRC-32 / Adler-32 of earlier segments of the same stream
    uint64_t total_before;  // bytes of earlier segments (gzip ISIZE)
    uint32_t *check_out;  // per unit: running check after this segment (may be null)
};

// LDS bit buffer -> HBM.  `nbits` valid bits in obuf; writes the complete bytes (all of them if
// `all`), keeps the rest at the front.  Returns bytes written.
template <class LDS>
__device__ uint32_t flush_bits(LDS &L, uint8_t *gout, uint32_t cap, uint32_t obytes, uint32_t &nbits, bool all)
{
    // (LSYNC, not WSYNC: only LDS is handed between lanes here; a full fence would wait for the candidate loads of the
    // next chunk and for the stores just issued)
    LSYNC();
    const uint32_t lane = lane_id();
    uint32_t nbytes = all ? (nbits + 7u) >> 3 : (nbits >> 5) << 2;  // whole dwords unless closing
    const uint8_t *src = (const uint8_t *)L.obuf;
    // whole dwords (gfx950 stores them at any address), then the last bytes
    struct __attribute__((packed, aligned(1))) U32u {
        uint32_t v;
    };
    const uint32_t ndw = nbytes >> 2;
    for (uint32_t j = lane; j < ndw; j += 64) {
        const uint32_t at = obytes + 4u * j;
        if (at + 4u <= cap) {
            ((U32u *)(gout + at))->v = L.obuf[j];
        } else {
            for (uint32_t b = 0; b < 4; b++)
                if (at + b < cap) gout[at + b] = src[4u * j + b];
        }
    }
    for (uint32_t j = 4u * ndw + lane; j < nbytes; j += 64)
        if (obytes + j < cap) gout[obytes + j] = src[j];
    LSYNC();
    // keep the partial dword, clear the rest
    uint32_t keep_w = nbytes >> 2;
    uint32_t carry = (!all && keep_w < LDS::OBUF) ? L.obuf[keep_w] : 0u;
    LSYNC();
    for (uint32_t j = lane; j < LDS::OBUF; j += 64) L.obuf[j] = j == 0 ? carry : 0u;
    LSYNC();
    nbits = all ? 0u : nbits - nbytes * 8u;
    return nbytes;
}

// append `n` bits (uniform value) to the bit buffer
template <class LDS>
__device__ __forceinline__ void put_uniform(LDS &L, uint32_t &nbits, uint32_t bits, uint32_t n)
{
    if (lane_id() == 0 && n) {
        uint32_t w = nbits >> 5, sh = nbits & 31u;
        atomicOr(&L.obuf[w], bits << sh);
        if (sh && (n + sh > 32)) atomicOr(&L.obuf[w + 1], bits >> (32 - sh));
    }
    nbits += n;
}

__device__ __forceinline__ uint32_t stored_size(uint32_t n, bool sync)
{
    uint32_t blocks = n ? (n + 65534u) / 65535u : 1u;
    return n + 5u * blocks + (sync ? 5u : 0u);
}

// ---- dynamic Huffman blocks (levels 2..9): oracle/oracle_deflate.c write_block() ------------------------------------
constexpr uint32_t TOK_BLOCK = 65536;  // tokens per block (a 64 KiB unit is one block: the per-block work, three code
                                       // constructions and a second pass over the tokens, is paid once); a block closes once a
                                       // further chunk (64 tokens) might not fit
constexpr uint32_t TOK_MATCH = 0x80000000u;
__device__ __constant__ static const uint8_t CL_ORDER[20] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 0};

// WAYS = positions kept per hash slot: 1 (levels 2..5), 2 (levels 6..9: the newest two; table[0] newest)
template <int WAYS_>
struct alignas(16) DLdsT {
    static constexpr int WAYS = WAYS_;
    static constexpr uint32_t OBUF = 416;  // dwords; a group of 64 tokens adds at most 96
    uint16_t table[WAYS_ << HASH_BITS];
    uint32_t obuf[OBUF];
    uint32_t lfreq[288], dfreq[32], cfreq[20];
    union {
        struct {
            uint32_t key[288];     // (frequency << 9 | symbol) of every symbol, unsorted: read by all lanes to rank their own
            uint32_t w[576];       // leaves (sorted) then internal nodes: weight, later depth
            uint16_t parent[576];
            uint16_t sym[288];     // symbol of sorted leaf i
        } tree;
        struct {
            uint32_t lcode[288], dcode[32], ccode[20];  // bit-reversed code | length << 16
        } code;
    };
    uint16_t seq[320];  // code-length sequence: symbol | extra << 8
    uint8_t ll[288], dl[32], cl[20];
    uint32_t lentab[256];  // per match length 3..258: length code 0..28 | extra-bit count << 8 | extra-bit value << 16
};
using DLds = DLdsT<1>;
using DLds2 = DLdsT<2>;
static_assert(sizeof(DLds) <= 19200, "eight waves per CU");
static_assert(sizeof(DLds2) <= 26880, "six waves per CU");

// Code lengths of freq[0..n) (n <= 288) limited to maxbits into len[0..n): Huffman over (frequency, symbol)-sorted
// leaves, two-queue merge with ties to the leaf; too deep -> all frequencies halved (rounding up) and rebuilt.
// Sorting is a rank count (all lanes), the merge runs on lane 0, depths come from pointer jumping (all lanes).
template <class DL>
__device__ __forceinline__ void build_lengths(DL &L, const uint32_t *freq, uint32_t n, uint32_t maxbits, uint8_t *len)
{
    const uint32_t lane = lane_id();
    uint32_t f[5], m = 0;
#pragma unroll
    for (uint32_t j = 0; j < 5; j++) {
        const uint32_t s = lane + 64u * j;
        f[j] = s < n ? freq[s] : 0u;
        m += (uint32_t)__popcll(__ballot(f[j] != 0));
        if (s < n) len[s] = 0;
    }
    if (m == 0) return;
    if (m == 1) {
#pragma unroll
        for (uint32_t j = 0; j < 5; j++)
            if (f[j]) len[lane + 64u * j] = 1;
        return;
    }
    const uint32_t root = 2u * m - 2u;
    for (;;) {
        uint32_t key[5], rank[5];
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) {
            const uint32_t s = lane + 64u * j;
            key[j] = f[j] ? (f[j] << 9) | s : 0xffffffffu;
            rank[j] = 0;
            if (s < n) L.tree.key[s] = key[j];
        }
        WSYNC();
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t k = L.tree.key[i];  // the same address in all lanes
#pragma unroll
            for (uint32_t j = 0; j < 5; j++) rank[j] += k < key[j] ? 1u : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < 5; j++)
            if (f[j]) {
                L.tree.w[rank[j]] = f[j];
                L.tree.sym[rank[j]] = (uint16_t)(lane + 64u * j);
            }
        WSYNC();
        if (lane == 0) {
            const uint32_t INF = 0xffffffffu;
            uint32_t li = 0, ii = m, nn = m, wl = L.tree.w[0], wi = INF;
            while (nn <= root) {
                uint32_t sum = 0;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    if (wl <= wi) {
                        sum += wl;
                        L.tree.parent[li] = (uint16_t)nn;
                        li++;
                        wl = li < m ? L.tree.w[li] : INF;
                    } else {
                        sum += wi;
                        L.tree.parent[ii] = (uint16_t)nn;
                        ii++;
                        wi = ii < nn ? L.tree.w[ii] : INF;
                    }
                }
                L.tree.w[nn] = sum;
                if (ii == nn) wi = sum;  // the internal queue was empty: the new node is its front
                nn++;
            }
            L.tree.parent[root] = (uint16_t)root;
        }
        WSYNC();
        // depth by pointer jumping: after round r a node knows min(depth, 2^r) and its ancestor that far up
        uint32_t d[9], pp[9];
#pragma unroll
        for (uint32_t j = 0; j < 9; j++) {
            const uint32_t i = lane + 64u * j;
            pp[j] = i <= root ? L.tree.parent[i] : root;
            d[j] = i < root ? 1u : 0u;
        }
        WSYNC();
        bool open = true;
        for (uint32_t r = 0; r < 5 && open; r++) {
#pragma unroll
            for (uint32_t j = 0; j < 9; j++) {
                const uint32_t i = lane + 64u * j;
                if (i <= root) {
                    L.tree.w[i] = d[j];
                    L.tree.parent[i] = (uint16_t)pp[j];
                }
            }
            WSYNC();
            bool mine_open = false;
#pragma unroll
            for (uint32_t j = 0; j < 9; j++) {
                const uint32_t i = lane + 64u * j;
                if (i <= root) {
                    d[j] += L.tree.w[pp[j]];
                    pp[j] = L.tree.parent[pp[j]];
                    mine_open |= pp[j] != root;
                }
            }
            WSYNC();
            open = __ballot(mine_open) != 0;
        }
        uint32_t deepest = 0;
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) {  // leaves are the first m nodes
            const uint32_t i = lane + 64u * j;
            if (i < m && d[j] > deepest) deepest = d[j];
        }
        const bool too_deep = open || __ballot(deepest > maxbits) != 0;
        if (!too_deep) {
#pragma unroll
            for (uint32_t j = 0; j < 5; j++) {
                const uint32_t i = lane + 64u * j;
                if (i < m) len[L.tree.sym[i]] = (uint8_t)d[j];
            }
            WSYNC();
            return;
        }
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) f[j] = f[j] ? (f[j] + 1u) >> 1 : 0u;
    }
}

// canonical codes (RFC 1951 sec. 3.2.2) of len[0..n) into code[0..n): bit-reversed code | length << 16
__device__ __forceinline__ void canon_codes(const uint8_t *len, uint32_t n, uint32_t *code)
{
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    uint32_t l[5], c[5];
#pragma unroll
    for (uint32_t j = 0; j < 5; j++) {
        const uint32_t s = lane + 64u * j;
        l[j] = s < n ? len[s] : 0u;
        c[j] = 0;
    }
    uint32_t base = 0;
    for (uint32_t b = 1; b <= 15; b++) {
        uint32_t running = 0;
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) {
            const uint64_t mm = __ballot(l[j] == b);
            if (l[j] == b) c[j] = base + running + (uint32_t)__popcll(mm & lt);
            running += (uint32_t)__popcll(mm);
        }
        base = (base + running) << 1;
    }
#pragma unroll
    for (uint32_t j = 0; j < 5; j++) {
        const uint32_t s = lane + 64u * j;
        if (s < n) code[s] = l[j] ? (__brev(c[j]) >> (32u - l[j])) | (l[j] << 16) : 0u;
    }
}

// one value of nb <= 48 bits per lane (nb = 0: none), appended in lane order
template <class DL>
__device__ __forceinline__ void put_lanes(DL &L, uint32_t &nbits, uint64_t bits, uint32_t nb)
{
    const uint32_t incl = wave_incl_scan(nb);
    if (nb) {
        const uint32_t at = nbits + incl - nb, w = at >> 5, sh = at & 31u;
        const uint64_t lo = bits << sh;
        atomicOr(&L.obuf[w], (uint32_t)lo);
        if (sh + nb > 32) atomicOr(&L.obuf[w + 1], (uint32_t)(lo >> 32));
        if (sh + nb > 64) atomicOr(&L.obuf[w + 2], (uint32_t)(bits >> (64u - sh)));
    }
    nbits += rdlane(incl, 63);
}

// bits of the stored form of `len` bytes when the stream stands at bit `at` (mod 8)
__device__ __forceinline__ uint32_t stored_bits(uint32_t at, uint32_t len)
{
    uint32_t p = at & 7u, off = 0;
    const uint32_t p0 = p;
    do {
        const uint32_t k = len - off < 65535u ? len - off : 65535u;
        p = ((p + 3u + 7u) & ~7u) + 32u + 8u * k;
        off += k;
    } while (off < len);
    return p - p0;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return rdlane(wave_incl_scan(v), 63); }

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// One unit (one wave).  DYN = false: level 0/1 and Z_FIXED, one fixed-Huffman block or stored blocks (LDS = ELds).
// DYN = true: levels 2..9, the chunks' tokens go to tokbuf (TOK_BLOCK words of HBM scratch of this wave) and every
// TOK_BLOCK-64.. tokens a block is written with the cheapest of dynamic / fixed / stored (LDS = DLds).
// Diagnostic build (-DCHIP_STATS, ESTATS=1 tools/time_encode.py): cycles per phase of the level-1 chunk loop, 24 words per unit.
#ifdef CHIP_STATS
#define ET_BEGIN(v) const unsigned long long v = __builtin_readcyclecounter()
#define ET_END(i, v) (est[i] += __builtin_readcyclecounter() - (v))
#define EC(i, n) (est[i] += (unsigned long long)(n))
#else
#define ET_BEGIN(v)
#define ET_END(i, v)
#define EC(i, n)
#endif

template <bool DYN, class LDS>
__device__ __forceinline__ void encode_unit(const EncArgs &a, const uint32_t u, LDS &L, uint32_t *tokbuf)
{
    const uint32_t lane = lane_id();
#ifdef CHIP_STATS
    unsigned long long est[24] = {};
    const unsigned long long et0 = __builtin_readcyclecounter();
#endif
    const uint8_t *gin = a.b.in_base + a.b.in_off[u];
    const uint32_t n = a.b.in_len[u];
    uint8_t *gout = a.b.out_base + a.b.out_off[u];
    const uint32_t cap = a.b.out_cap[u];
    const bool hdr = a.flags & 1u, trl = a.flags & 2u, final = a.flags & 4u;
    const uint32_t strategy = (a.flags >> 8) & 7u;
    const bool no_match = strategy == CHIP_STRATEGY_HUFFMAN_ONLY, rle = strategy == CHIP_STRATEGY_RLE;
    const bool lazy = a.level >= 4;  // (dynamic levels only: zlib's lazy matching starts at level 4 too)
    const int fmt = a.b.format;

    const uint32_t mis = (uint32_t)((uintptr_t)gin & 3u);
    const uint32_t *g32 = (const uint32_t *)(gin - mis);
    const uint32_t total_dw = (mis + n + 3u) >> 2;

    for (uint32_t j = lane; j < (uint32_t)(LDS::WAYS << HASH_BITS) / 2; j += 64) ((uint32_t *)L.table)[j] = 0;
    for (uint32_t j = lane; j < LDS::OBUF; j += 64) L.obuf[j] = 0;
    for (uint32_t j = lane; j < 256; j += 64) {
        uint32_t lc, lext, lxv;
        len_parts(j + 3u, lc, lext, lxv);
        if constexpr (DYN) {
            L.lentab[j] = lc | (lext << 8) | (lxv << 16);
        } else {
            const uint32_t sym = 257u + lc, nb = sym < 280 ? 7u : 8u;
            const uint32_t code = sym < 280 ? rev_bits(sym - 256, 7) : rev_bits(0xC0 + (sym - 280), 8);
            L.lentab[j] = code | (lxv << nb) | ((nb + lext) << 16);
        }
    }
    if constexpr (DYN) {
        for (uint32_t j = lane; j < 288; j += 64) L.lfreq[j] = 0;
        if (lane < 32) L.dfreq[lane] = 0;
        if (lane < 20) L.cfreq[lane] = 0;
    }
    WSYNC();

    uint32_t obytes = 0, nbits = 0;
    // ---- wrapper header ---------------------------------------------------------------------------
    if (hdr && fmt == CHIP_FMT_GZIP) {
        // RFC 1952 sec. 2.3: no name / time; XFL 4 = fastest (level 1), 2 = best (level 9); OS 3 = Unix
        put_uniform(L, nbits, 0x00088b1fu, 32);
        put_uniform(L, nbits, 0, 32);
        put_uniform(L, nbits, (a.level == 9 ? 2u : a.level == 1 ? 4u : 0u) | (3u << 8), 16);
    } else if (hdr && fmt == CHIP_FMT_ZLIB) {
        uint32_t flevel = a.level < 2 ? 0u : a.level < 6 ? 1u : a.level == 6 ? 2u : 3u;
        uint32_t h = (0x78u << 8) | (flevel << 6);
        h += 31u - h % 31u;
        put_uniform(L, nbits, (h >> 8) | ((h & 0xffu) << 8), 16);
    }
    const uint32_t hdr_bytes = nbits >> 3;
    const uint32_t ssz = stored_size(n, !final);
    bool use_stored = a.level == 0;
    uint32_t body_bytes = 0;

    if (!use_stored) {
        if constexpr (!DYN) put_uniform(L, nbits, (final ? 1u : 0u) | (1u << 1), 3);
        uint32_t skip = 0;
        uint32_t ntok = 0, from = 0, xb_lane = 0, nm_lane = 0;  // DYN: the open block
        // Two chunks are in flight: while chunk c is measured, chosen and emitted, chunk c+1 has already done its
        // table lookups and updates (they depend on hashes and positions only, never on what matched) and its
        // candidate bytes are on their way from memory.  Every lane issues the same eleven loads per chunk (lanes
        // without a candidate compare their own position with itself) so that waiting for chunk c's bytes can leave
        // chunk c+1's in flight.
        struct Cand {
            uint32_t v, q;
            U128u qv, pv;  // 16 bytes at the candidate and at the position (lanes at least 16 bytes before the unit's end)
            bool has;
            uint32_t q2;   // two ways: the slot's older position
            U128u qv2;
            bool has2;
        };
        constexpr bool TWO = LDS::WAYS == 2;
        const uint32_t last_dw = total_dw ? total_dw - 1u : 0u;
        // 16-byte loads start at most here; a unit under 16 bytes reads the aligned 16 bytes around its start (same page) and ignores them
        const uint32_t wide_end = n >= 16 ? n - 16u : 0u;
        const uint8_t *gwide = gin - (n >= 16 ? 0u : (uint32_t)((uintptr_t)gin & 15u));
        // the 4 bytes at a lane's position are fetched one chunk ahead as two raw aligned dwords and put together when
        // the chunk is looked up (combining them at once would wait for the load, and for every load before it)
        uint32_t vd0, vd1;
        auto fetch_v = [&](uint32_t p) __attribute__((always_inline)) {
            const uint32_t i = (mis + p) >> 2;  // clamped into the unit
            vd0 = g32[i < last_dw ? i : last_dw];
            vd1 = g32[i + 1 < last_dw ? i + 1 : last_dw];
        };
        fetch_v(lane);
        auto lookup = [&](Cand &c, uint32_t base) __attribute__((always_inline)) {
            const uint32_t p = base + lane;
            const bool valid4 = p + 4 <= n && base < n;
            {
                const uint32_t off = mis + p, i = off >> 2;
                c.v = p < n ? __builtin_amdgcn_alignbit(i + 1 < total_dw ? vd1 : 0u, i < total_dw ? vd0 : 0u, (off & 3u) * 8u) : 0u;
            }
            fetch_v(p + 64u);
            // (selects, not an `if (valid4)` region: every exec-mask region costs the scalar unit three instructions and a branch, and this
            // loop has a dozen of them per chunk; a lane without four bytes hashes the zero it holds and reads a slot it ignores)
            const uint32_t h = (c.v * 2654435761u) >> (32 - HASH_BITS);
            const uint32_t old0 = L.table[h];
            const uint32_t dist = (p - old0) & 0xffffu;  // the nearest earlier position with the slot's low 16 bits
            c.has = valid4 && dist - 1u < MAX_DIST && dist <= p;
            c.q = c.has ? p - dist : p;
            c.has2 = false;
            c.q2 = p;
            if constexpr (TWO) {
                const uint32_t dist2 = (p - L.table[(1u << HASH_BITS) + h]) & 0xffffu;
                c.has2 = valid4 && dist2 - 1u < MAX_DIST && dist2 <= p && dist2 != dist;
                c.q2 = c.has2 ? p - dist2 : p;
            }
            // Z_RLE: the only candidate is the byte before (distance 1); Z_HUFFMAN_ONLY: none
            if (rle) {
                c.has = valid4 && p > 0;
                c.q = c.has ? p - 1 : p;
            }
            if (no_match) {
                c.has = false;
                c.q = p;
            }
            if (rle || no_match) {
                c.has2 = false;
                c.q2 = p;
            }
            // one unaligned 16-byte load per side: a scattered load costs the L1 a tag lookup per lane and instruction
            // (every lane issues both loads, clamped into the unit, so that waiting for chunk c's bytes leaves chunk
            // c+1's in flight; lanes in the unit's last 15 bytes do not use them)
#ifdef CHIP_EXP_NO_CAND  // traffic ablation (tools/profile_round.sh): no candidates at all -- every lane's candidate load reads its own
            c.has = c.has2 = false;  // position, nothing is measured, the unit becomes literals; what is left is the input's own stream
            c.q = c.q2 = p;
            c.qv = *(const U128u *)(gwide + (p < wide_end ? p : wide_end));
            if constexpr (TWO) c.qv2 = c.qv;
#else
            c.qv = *(const U128u *)(gwide + (c.q < wide_end ? c.q : wide_end));
            if constexpr (TWO) c.qv2 = *(const U128u *)(gwide + (c.q2 < wide_end ? c.q2 : wide_end));
#endif
            c.pv = *(const U128u *)(gwide + (p < wide_end ? p : wide_end));
            LSYNC();  // every lookup saw the table as it stood before this chunk (LDS only: the loads stay in flight)
            // The highest position of a slot stands.  (LDS has no 16-bit maximum: all write, a lane that finds a lower lane of
            // this chunk in its slot writes again -- a second round in one chunk of five, hardly ever a third.)
            if (valid4) L.table[h] = (uint16_t)p;
            for (;;) {
                LSYNC();
                const bool lost = valid4 && ((L.table[h] - base) & 0xffffu) < lane;
                if (!__any(lost)) break;
                if (lost) L.table[h] = (uint16_t)p;
            }
            if constexpr (TWO) {
                // the older way: the second highest position of the chunk in the slot, or what the newer way held before the chunk.
                // The slot's winner puts the old value there, then the others settle their highest as above.
                const bool winner = valid4 && ((L.table[h] - base) & 0xffffu) == lane;
                uint16_t *const t2 = L.table + (1u << HASH_BITS);
                if (winner) t2[h] = (uint16_t)old0;
                LSYNC();
                const bool second = valid4 && !winner;
                if (second) t2[h] = (uint16_t)p;
                for (;;) {
                    LSYNC();
                    const bool lost2 = second && ((t2[h] - base) & 0xffffu) < lane;  // (what a loser reads here a loser of this chunk wrote)
                    if (!__any(lost2)) break;
                    if (lost2) t2[h] = (uint16_t)p;
                }
            }
        };
        // one chunk: look the next one up (its loads fly while this one is worked on), measure, choose, emit
        auto step = [&](Cand &cur, Cand &nxt, const uint32_t base) __attribute__((always_inline)) {
            ET_BEGIN(et1);
            lookup(nxt, base + 64);  // past the end this is an empty chunk: same loads, nothing looked up
            ET_END(1, et1);
            EC(8, 1);
            ET_BEGIN(et2);
            const uint32_t p = base + lane;
            uint32_t mlen = 0, mdist = 0;
            const uint32_t v = cur.v;
            // common prefix of the position with a candidate (its first 16 bytes in qv), 0 if under MIN_MATCH
            auto measure = [&](const bool has, const uint32_t q, const U128u &qv) __attribute__((always_inline)) {
                const uint32_t lim = n - p < MAX_MATCH ? n - p : MAX_MATCH;
                // the first 16 bytes of both sides arrived together: most candidates are decided right here, without a branch; the
                // loop below runs only when some lane's candidate is not (a uniform test), and only for those lanes
                const bool wide = p + 16 <= n;
                uint32_t k = wide ? first_diff16(qv, cur.pv) : 0u;
                const bool more = has && (!wide || k >= 16u) && k < lim;
                if (__any(more)) {
                    if (more) {
                        while (k < lim) {
                            if (p + k + 16 <= n) {
                                const uint32_t d = first_diff16(*(const U128u *)(gin + q + k), *(const U128u *)(gin + p + k));
                                k += d;
                                if (d < 16) break;
                            } else {
                                uint32_t x = ld32(g32, total_dw, mis + q + k) ^ ld32(g32, total_dw, mis + p + k);
                                if (x) {
                                    k += ((uint32_t)__ffs((int)x) - 1u) >> 3;
                                    break;
                                }
                                k += 4;
                            }
                        }
                    }
                }
                if (k > lim) k = lim;
                return has && k >= MIN_MATCH ? k : 0u;
            };
            mlen = measure(cur.has, cur.q, cur.qv);
            mdist = mlen ? p - cur.q : 0u;
            if constexpr (TWO) {
                const uint32_t k2 = measure(cur.has2, cur.q2, cur.qv2);  // the older position wins only with a longer match
                mdist = k2 > mlen ? p - cur.q2 : mdist;
                mlen = k2 > mlen ? k2 : mlen;
            }
            // greedy choice, left to right over the chunk: jump from selected match to selected match (a scalar
            // step per chosen match, not per position); everything in between is a literal
            const uint32_t lim64 = n - base < 64 ? n - base : 64;
            const uint64_t limmask = lim64 >= 64 ? ~0ull : ((1ull << lim64) - 1ull);
            ET_END(2, et2);
            ET_BEGIN(et3);
            const uint64_t cand = __ballot(mlen >= MIN_MATCH) & limmask;
            EC(9, __popcll(cand));
            uint64_t covered = 0;  // positions inside a chosen match (its start excluded)
            uint64_t deferred = 0;  // lazy levels: match candidates that gave way to the next position and became literals
            uint32_t pos = rdfirst(skip);
            {
                const uint32_t lim_s = rdfirst(lim64);
                const uint64_t cand_s = ((uint64_t)rdfirst((uint32_t)(cand >> 32)) << 32) | rdfirst((uint32_t)cand);
                // while (pos < lim64) { rest = cand & (~0 << pos); if (!rest) break; c = ctz(rest); len = mlen of lane c;
                //   pos = c + len; covered |= min(len - 1, 63 - c) ones from bit c + 1; }   -- 14 scalar instructions per chosen
                // match (the compiler's version of the same loop: 19)
                uint64_t t, m;
                uint32_t c, len, sz, room;
                if (DYN && lazy) {
                    // levels 4..9: a match gives way to a longer one at the next position of the chunk (it becomes a literal)
                    uint32_t len1;
                    asm volatile(
                        "1:\n\t"
                        "s_cmp_ge_u32 %[pos], %[lim]\n\t"
                        "s_cbranch_scc1 3f\n\t"
                        "s_lshl_b64 %[t], -1, %[pos]\n\t"
                        "s_and_b64 %[t], %[t], %[cand]\n\t"
                        "s_cbranch_scc0 3f\n\t"
                        "s_ff1_i32_b64 %[c], %[t]\n\t"
                        "v_readlane_b32 %[len], %[mlen], %[c]\n\t"
                        "s_add_u32 %[sz], %[c], 1\n\t"
                        "s_cmp_ge_u32 %[sz], %[lim]\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "v_readlane_b32 %[len1], %[mlen], %[sz]\n\t"
                        "s_cmp_gt_u32 %[len1], %[len]\n\t"
                        "s_cbranch_scc0 2f\n\t"
                        "s_lshl_b64 %[t], 1, %[c]\n\t"
                        "s_or_b64 %[def], %[def], %[t]\n\t"
                        "s_mov_b32 %[pos], %[sz]\n\t"
                        "s_branch 1b\n\t"
                        "2:\n\t"
                        "s_add_u32 %[pos], %[c], %[len]\n\t"
                        "s_xor_b32 %[room], %[c], 63\n\t"
                        "s_add_u32 %[sz], %[len], -1\n\t"
                        "s_min_u32 %[sz], %[sz], %[room]\n\t"
                        "s_add_u32 %[c], %[c], 1\n\t"
                        "s_bfm_b64 %[m], %[sz], %[c]\n\t"
                        "s_or_b64 %[cov], %[cov], %[m]\n\t"
                        "s_branch 1b\n\t"
                        "3:"
                        : [pos] "+s"(pos), [cov] "+s"(covered), [t] "=&s"(t), [m] "=&s"(m), [c] "=&s"(c), [len] "=&s"(len), [sz] "=&s"(sz), [room] "=&s"(room),
                          [len1] "=&s"(len1), [def] "+s"(deferred)
                        : [lim] "s"(lim_s), [cand] "s"(cand_s), [mlen] "v"(mlen)
                        : "scc");
                } else {
                asm volatile(
                    "1:\n\t"
                    "s_cmp_ge_u32 %[pos], %[lim]\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "s_lshl_b64 %[t], -1, %[pos]\n\t"
                    "s_and_b64 %[t], %[t], %[cand]\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "s_ff1_i32_b64 %[c], %[t]\n\t"
                    "v_readlane_b32 %[len], %[mlen], %[c]\n\t"
                    "s_add_u32 %[pos], %[c], %[len]\n\t"
                    "s_xor_b32 %[room], %[c], 63\n\t"
                    "s_add_u32 %[sz], %[len], -1\n\t"
                    "s_min_u32 %[sz], %[sz], %[room]\n\t"
                    "s_add_u32 %[c], %[c], 1\n\t"
                    "s_bfm_b64 %[m], %[sz], %[c]\n\t"
                    "s_or_b64 %[cov], %[cov], %[m]\n\t"
                    "s_branch 1b\n\t"
                    "2:"
                    : [pos] "+s"(pos), [cov] "+s"(covered), [t] "=&s"(t), [m] "=&s"(m), [c] "=&s"(c), [len] "=&s"(len), [sz] "=&s"(sz), [room] "=&s"(room)
                    : [lim] "s"(lim_s), [cand] "s"(cand_s), [mlen] "v"(mlen)
                    : "scc");
                }
            }
            ET_END(3, et3);
            ET_BEGIN(et4);
            const uint64_t sel = skip < 64 ? limmask & ~covered & (~0ull << skip) : 0ull;
            EC(10, __popcll(sel));
            skip = pos > 64 ? pos - 64 : 0;
            const bool mine = (sel >> lane) & 1ull;
            if constexpr (!DYN) {
                // literal and match both computed by every lane, one kept, a lane that emits nothing adds zero bits: no exec-mask regions
                uint32_t nb, bits;
                {
                    const bool is_match = mlen >= MIN_MATCH;
                    const uint32_t lit = v & 0xffu, lnb = lit < 144 ? 8u : 9u;
                    const uint32_t lbits = rev_bits(lit < 144 ? 0x30u + lit : 0x100u + lit, 9) >> (9u - lnb);
                    const uint32_t e = L.lentab[is_match ? mlen - 3u : 0u];
                    uint32_t dc, dext, dxv;
                    dist_parts(is_match ? mdist : 1u, dc, dext, dxv);
                    uint32_t mnb = e >> 16, mbits = e & 0xffffu;
                    mbits |= rev_bits(dc, 5) << mnb;
                    mnb += 5;
                    mbits |= dxv << mnb;
                    mnb += dext;
                    bits = is_match ? mbits : lbits;
                    nb = is_match ? mnb : lnb;
                    nb = mine ? nb : 0u;
                    bits = mine ? bits : 0u;
                }
                const uint32_t incl = wave_incl_scan(nb);
                {
                    const uint32_t at = nbits + incl - nb, w = at >> 5, sh = at & 31u;
                    atomicOr(&L.obuf[w], bits << sh);  // (zero for a lane that emits nothing; its word lies inside the buffer)
                    atomicOr(&L.obuf[w + 1], (sh && nb + sh > 32) ? bits >> (32 - sh) : 0u);
                }
                nbits += rdlane(incl, 63);
                ET_END(4, et4);
                ET_BEGIN(et5);
                if (nbits > (uint32_t)(OUT_DW - 64) * 32u) obytes += flush_bits(L, gout, cap, obytes, nbits, false);
                ET_END(5, et5);
            } else {
                // tokens to the scratch in order, symbol counts to LDS
                if (mine) {
                    uint32_t t = v & 0xffu;
                    if (mlen >= MIN_MATCH && !((deferred >> lane) & 1ull)) {
                        uint32_t dc, dext, dxv;
                        const uint32_t e = L.lentab[mlen - 3u];
                        dist_parts(mdist, dc, dext, dxv);
                        t = TOK_MATCH | ((mdist - 1u) << 9) | (mlen - 3u);
                        atomicAdd(&L.lfreq[257u + (e & 0xffu)], 1u);
                        atomicAdd(&L.dfreq[dc], 1u);
                        xb_lane += ((e >> 8) & 0xffu) + dext;
                        nm_lane += 1;
                    } else {
                        atomicAdd(&L.lfreq[t], 1u);
                    }
                    tokbuf[ntok + (uint32_t)__popcll(sel & lanemask_lt())] = t;
                }
                ntok += (uint32_t)__popcll(sel);
            }
        };
        // Two chunks per trip with the candidate registers changing roles, so that no loaded register has to be copied
        // (a copy would wait for the loads just issued).
        Cand A, B;
        lookup(A, 0);
        for (uint32_t base = 0; base < n || (DYN && base == 0); base += 64) {
            step(A, B, base);
            bool ended = false, close = false;
            if constexpr (DYN) {
                ended = base + 64 >= n;
                close = ended || ntok > TOK_BLOCK - 64;
            }
            if (!close && base + 64 < n) {
                base += 64;
                step(B, A, base);
                if constexpr (DYN) {
                    ended = base + 64 >= n;
                    close = ended || ntok > TOK_BLOCK - 64;
                }
            } else {
                A = B;
            }
            if constexpr (DYN) {
                if (close) {
                    const uint32_t to = ended ? n : base + 64 + skip;
                    const bool lastb = final && ended;
                    const uint32_t xbits = wave_sum(xb_lane), nmatch = wave_sum(nm_lane);
                    if (lane == 0) L.lfreq[256] += 1;
                    WSYNC();
                    build_lengths(L, L.lfreq, 286, 15, L.ll);
                    build_lengths(L, L.dfreq, 30, 15, L.dl);
                    if (lane == 0) {
                        L.ll[286] = L.ll[287] = 0;
                        L.dl[30] = L.dl[31] = 0;
                        if (!nmatch) L.dl[0] = 1;  // one distance code is always described
                    }
                    WSYNC();
                    uint32_t hl = 257, hd = 1;
#pragma unroll
                    for (uint32_t j = 0; j < 5; j++) {
                        const uint32_t sy = lane + 64u * j;
                        if (sy < 286 && L.ll[sy] && sy + 1 > hl) hl = sy + 1;
                    }
                    if (lane < 30 && L.dl[lane]) hd = lane + 1;
                    const uint32_t hlit = wave_max(hl), hdist = wave_max(hd);
                    // code-length sequence with greedy runs (lane 0)
                    uint32_t ns = 0, cl_extra = 0;
                    if (lane == 0) {
                        const uint32_t na = hlit + hdist;
                        uint32_t i = 0;
                        uint32_t nxt_v = L.ll[0];
                        while (i < na) {
                            const uint32_t val = nxt_v;
                            uint32_t run = 1;
                            for (;;) {
                                const uint32_t k = i + run;
                                if (k >= na) break;
                                nxt_v = k < hlit ? L.ll[k] : L.dl[k - hlit];
                                if (nxt_v != val) break;
                                run++;
                            }
                            i += run;
                            if (val == 0) {
                                while (run >= 11) {
                                    const uint32_t r = run < 138 ? run : 138;
                                    L.seq[ns++] = (uint16_t)(18u | ((r - 11u) << 8));
                                    L.cfreq[18] += 1;
                                    cl_extra += 7;
                                    run -= r;
                                }
                                if (run >= 3) {
                                    L.seq[ns++] = (uint16_t)(17u | ((run - 3u) << 8));
                                    L.cfreq[17] += 1;
                                    cl_extra += 3;
                                    run = 0;
                                }
                                L.cfreq[0] += run;
                                while (run-- > 0) L.seq[ns++] = 0;
                            } else {
                                L.seq[ns++] = (uint16_t)val;
                                L.cfreq[val] += 1;
                                run--;
                                while (run >= 3) {
                                    const uint32_t r = run < 6 ? run : 6;
                                    L.seq[ns++] = (uint16_t)(16u | ((r - 3u) << 8));
                                    L.cfreq[16] += 1;
                                    cl_extra += 2;
                                    run -= r;
                                }
                                L.cfreq[val] += run;
                                while (run-- > 0) L.seq[ns++] = (uint16_t)val;
                            }
                        }
                    }
                    ns = rdfirst(ns);
                    cl_extra = rdfirst(cl_extra);
                    WSYNC();
                    build_lengths(L, L.cfreq, 19, 7, L.cl);
                    uint32_t hc = 4;
                    if (lane < 19 && L.cl[CL_ORDER[lane < 19 ? lane : 19]]) hc = lane + 1 > 4 ? lane + 1 : 4;
                    const uint32_t hclen = wave_max(hc);
                    // costs in bits
                    uint32_t dynp = 0, fixp = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 5; j++) {
                        const uint32_t sy = lane + 64u * j;
                        if (sy < 286) {
                            const uint32_t fr = L.lfreq[sy];
                            dynp += fr * L.ll[sy];
                            fixp += fr * (sy < 144 ? 8u : sy < 256 ? 9u : sy < 280 ? 7u : 8u);
                        }
                    }
                    if (lane < 30) {
                        const uint32_t fr = L.dfreq[lane];
                        dynp += fr * L.dl[lane];
                        fixp += fr * 5u;
                    }
                    if (lane < 19) dynp += L.cfreq[lane] * L.cl[lane];
                    const uint32_t dyn = wave_sum(dynp) + 3u + 14u + 3u * hclen + cl_extra + xbits, fix = wave_sum(fixp) + 3u + xbits;
                    const bool use_dyn = dyn < fix;
                    const uint32_t huff = use_dyn ? dyn : fix;
                    WSYNC();
                    if (stored_bits(nbits, to - from) < huff) {
                        uint32_t off = from;
                        do {
                            const uint32_t k = to - off < 65535u ? to - off : 65535u;
                            put_uniform(L, nbits, (lastb && off + k == to) ? 1u : 0u, 3);
                            nbits = (nbits + 7u) & ~7u;
                            put_uniform(L, nbits, k | ((~k & 0xffffu) << 16), 32);
                            obytes += flush_bits(L, gout, cap, obytes, nbits, true);
                            for (uint32_t j = lane; j < k; j += 64)
                                if (obytes + j < cap) gout[obytes + j] = gin[off + j];
                            obytes += k;
                            off += k;
                        } while (off < to);
                    } else {
                        if (!use_dyn) {
                            for (uint32_t j = lane; j < 288; j += 64) L.ll[j] = j < 144 ? 8 : j < 256 ? 9 : j < 280 ? 7 : 8;
                            if (lane < 32) L.dl[lane] = lane < 30 ? 5 : 0;
                            WSYNC();
                        }
                        canon_codes(L.ll, 288, L.code.lcode);
                        canon_codes(L.dl, 30, L.code.dcode);
                        if (use_dyn) canon_codes(L.cl, 19, L.code.ccode);
                        WSYNC();
                        put_uniform(L, nbits, (lastb ? 1u : 0u) | ((use_dyn ? 2u : 1u) << 1), 3);
                        if (use_dyn) {
                            put_uniform(L, nbits, (hlit - 257u) | ((hdist - 1u) << 5) | ((hclen - 4u) << 10), 14);
                            put_lanes(L, nbits, lane < hclen ? L.cl[CL_ORDER[lane < 19 ? lane : 19]] : 0u, lane < hclen ? 3u : 0u);
                            for (uint32_t g = 0; g < ns; g += 64) {
                                uint32_t nb = 0;
                                uint64_t bits = 0;
                                if (g + lane < ns) {
                                    const uint32_t e = L.seq[g + lane], sy = e & 0xffu, ce = L.code.ccode[sy];
                                    nb = ce >> 16;
                                    bits = (ce & 0xffffu) | ((uint64_t)(e >> 8) << nb);
                                    nb += sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u;
                                }
                                put_lanes(L, nbits, bits, nb);
                                if (nbits > (DLds::OBUF - 104u) * 32u) obytes += flush_bits(L, gout, cap, obytes, nbits, false);
                            }
                        }
                        for (uint32_t g = 0; g < ntok; g += 64) {
                            uint32_t nb = 0;
                            uint64_t bits = 0;
                            if (g + lane < ntok) {
                                const uint32_t t = tokbuf[g + lane];
                                if (t & TOK_MATCH) {
                                    uint32_t dc, dext, dxv;
                                    const uint32_t e = L.lentab[t & 0xffu], lc = e & 0xffu, lext = (e >> 8) & 0xffu, lxv = e >> 16;
                                    dist_parts(((t >> 9) & 0x7fffu) + 1u, dc, dext, dxv);
                                    const uint32_t le = L.code.lcode[257u + lc], de = L.code.dcode[dc];
                                    bits = le & 0xffffu;
                                    nb = le >> 16;
                                    bits |= (uint64_t)lxv << nb;
                                    nb += lext;
                                    bits |= (uint64_t)(de & 0xffffu) << nb;
                                    nb += de >> 16;
                                    bits |= (uint64_t)dxv << nb;
                                    nb += dext;
                                } else {
                                    const uint32_t le = L.code.lcode[t];
                                    bits = le & 0xffffu;
                                    nb = le >> 16;
                                }
                            }
                            put_lanes(L, nbits, bits, nb);
                            if (nbits > (DLds::OBUF - 104u) * 32u) obytes += flush_bits(L, gout, cap, obytes, nbits, false);
                        }
                        const uint32_t eob = L.code.lcode[256];
                        put_uniform(L, nbits, eob & 0xffffu, eob >> 16);
                    }
                    // the next block starts empty
                    WSYNC();
                    for (uint32_t j = lane; j < 288; j += 64) L.lfreq[j] = 0;
                    if (lane < 32) L.dfreq[lane] = 0;
                    if (lane < 20) L.cfreq[lane] = 0;
                    WSYNC();
                    ntok = 0;
                    from = to;
                    xb_lane = nm_lane = 0;
                }
            }
        }
        if constexpr (!DYN) put_uniform(L, nbits, 0, 7);  // end of block
        if (!final) {
            put_uniform(L, nbits, 0, 3);  // empty stored block = sync marker (Z_SYNC_FLUSH)
            nbits = (nbits + 7u) & ~7u;
            put_uniform(L, nbits, 0xffff0000u, 32);
        } else nbits = (nbits + 7u) & ~7u;
        obytes += flush_bits(L, gout, cap, obytes, nbits, true);
        body_bytes = obytes - hdr_bytes;
        if (!DYN && ssz < body_bytes) use_stored = true;
    }
    if (use_stored) {
        // rewrite the body as stored blocks (RFC 1951 sec. 3.2.4); the header bytes are already in place
        if (a.level == 0) obytes += flush_bits(L, gout, cap, obytes, nbits, true);
        obytes = hdr_bytes;
        uint32_t off = 0;
        do {
            const uint32_t k = n - off < 65535u ? n - off : 65535u;
            const bool lastb = final && off + k == n;
            if (lane < 5) {
                const uint32_t hb[5] = {lastb ? 1u : 0u, k & 0xffu, k >> 8, (~k) & 0xffu, ((~k) >> 8) & 0xffu};
                if (obytes + lane < cap) gout[obytes + lane] = (uint8_t)hb[lane];
            }
            obytes += 5;
            for (uint32_t j = lane; j < k; j += 64)
                if (obytes + j < cap) gout[obytes + j] = gin[off + j];
            obytes += k;
            off += k;
        } while (off < n);
        if (!final) {
            if (lane < 5) {
                const uint32_t sb[5] = {0, 0, 0, 0xff, 0xff};
                if (obytes + lane < cap) gout[obytes + lane] = (uint8_t)sb[lane];
            }
            obytes += 5;
        }
    }
    // ---- checksum / trailer -----------------------------------------------------------------------
    uint32_t check = a.check_seed;
    if (fmt == CHIP_FMT_GZIP) check = wave_crc32((LDS_AS uint32_t *)L.table, gin, n, a.check_seed);  // the hash table is dead by now
    else if (fmt == CHIP_FMT_ZLIB) check = wave_adler32(gin, n, a.check_seed);
    if (trl && fmt == CHIP_FMT_GZIP) {
        const uint32_t isize = (uint32_t)(a.total_before + n);
        if (lane < 8) {
            uint32_t w = lane < 4 ? check : isize;
            if (obytes + lane < cap) gout[obytes + lane] = (uint8_t)(w >> (8 * (lane & 3u)));
        }
        obytes += 8;
    } else if (trl && fmt == CHIP_FMT_ZLIB) {
        if (lane < 4 && obytes + lane < cap) gout[obytes + lane] = (uint8_t)(check >> (8 * (3 - lane)));
        obytes += 4;
    }
    if (lane == 0) {
        a.b.out_len[u] = obytes <= cap ? obytes : cap;
        a.b.status[u] = obytes <= cap ? CHIP_ENC_FINISHED : CHIP_ENC_NEED_OUTPUT;
#ifdef CHIP_STATS
        if (a.b.stats) {
            est[0] = __builtin_readcyclecounter() - et0;
            for (int i = 0; i < 24; i++) a.b.stats[(size_t)u * 24 + i] = est[i];
        }
#endif
        if (a.b.in_used) a.b.in_used[u] = n;
        if (a.check_out) a.check_out[u] = check;
    }
}

__global__ __launch_bounds__(64) void deflate_kernel(EncArgs a)
{
    __shared__ ELds L;
    if (blockIdx.x >= a.b.n) return;
    encode_unit<false>(a, blockIdx.x, L, nullptr);
}

// Persistent grid (the token scratch is per resident wave): each wave takes the next unit from *next_unit.
template <class DL>
__device__ __forceinline__ void dyn_grid(const EncArgs &a, DL &L, uint32_t *scratch, uint32_t *next_unit)
{
    uint32_t *tokbuf = scratch + (size_t)blockIdx.x * TOK_BLOCK;
    for (;;) {
        uint32_t i = 0;
        if (lane_id() == 0) i = atomicAdd(next_unit, 1u);
        i = rdfirst(i);
        if (i >= a.b.n) break;
        encode_unit<true>(a, i, L, tokbuf);
        WSYNC();  // the next unit reuses the LDS
    }
}
__global__ __launch_bounds__(64) void deflate_dyn_kernel(EncArgs a, uint32_t *scratch, uint32_t *next_unit)
{
    __shared__ DLds L;
    dyn_grid(a, L, scratch, next_unit);
}
// levels 6..9: two positions per hash slot
__global__ __launch_bounds__(64) void deflate_dyn2_kernel(EncArgs a, uint32_t *scratch, uint32_t *next_unit)
{
    __shared__ DLds2 L;
    dyn_grid(a, L, scratch, next_unit);
}

// Token scratch and the unit counter of the dynamic-level launches, cached per (device, stream) like the
// inflate kernel's (inflate.hip slot_for): sized for min(n, resident waves) waves.
struct EncSlot {
    uint32_t *scratch = nullptr;
    uint32_t *counter = nullptr;
    int blocks = 0;
};
std::mutex g_enc_mu;
std::map<std::pair<int, hipStream_t>, EncSlot> g_enc_slots;

// (caller holds g_enc_mu)
hipError_t enc_slot_for(hipStream_t stream, uint32_t n, EncSlot &out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    EncSlot &sl = g_enc_slots[{dev, stream}];
    static int max_blocks[64] = {0};
    const int di = dev < 64 ? dev : 63;
    if (!max_blocks[di]) {
        int per_cu = 0, cus = 0;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, deflate_dyn_kernel, 64, 0)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        max_blocks[di] = per_cu * cus;
    }
    const int want = n < (uint32_t)max_blocks[di] ? (int)n : max_blocks[di];
    if (sl.blocks < want) {
        if (sl.scratch && (e = hipStreamSynchronize(stream)) != hipSuccess) return e;  // launches on the stream still use it
        (void)hipFree(sl.scratch);
        sl.scratch = nullptr;
        sl.blocks = 0;
        const int blocks = want <= 1 ? 1 : (want + want / 4 < max_blocks[di] ? want + want / 4 : max_blocks[di]);
        uint32_t *p = nullptr;
        if ((e = hipMalloc((void **)&p, (size_t)blocks * TOK_BLOCK * 4 + 256)) != hipSuccess) return e;
        sl.scratch = p;
        sl.counter = p + (size_t)blocks * TOK_BLOCK;
        sl.blocks = blocks;
    }
    out = sl;
    return hipSuccess;
}

}  // namespace

hipError_t release_deflate_scratch()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_enc_mu);
    for (auto it = g_enc_slots.begin(); it != g_enc_slots.end();) {
        if (it->first.first == dev) {
            (void)hipFree(it->second.scratch);
            it = g_enc_slots.erase(it);
        } else {
            ++it;
        }
    }
    return hipSuccess;
}

void release_deflate_scratch_of(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_enc_mu);
    auto it = g_enc_slots.find({dev, stream});
    if (it != g_enc_slots.end()) {
        (void)hipFree(it->second.scratch);
        g_enc_slots.erase(it);
    }
}

hipError_t launch_deflate_l1(const BatchArgs &b, int level, uint32_t flags, uint32_t check_seed, uint64_t total_before,
                             uint32_t *check_out, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    EncArgs a;
    a.b = b;
    a.level = level;
    a.flags = flags;
    a.check_seed = check_seed;
    a.total_before = total_before;
    a.check_out = check_out;
    const uint32_t strategy = (flags >> 8) & 7u;
    if (level >= 2 && strategy != CHIP_STRATEGY_FIXED) {
        // One lock from the slot's lookup to the launch (as launch_inflate): another host thread launching a larger batch on the same
        // stream may free and reallocate the scratch in enc_slot_for(); the counter reset and the kernel reach the stream back to back.
        std::lock_guard<std::mutex> lk(g_enc_mu);
        EncSlot sl;
        hipError_t e = enc_slot_for(stream, b.n, sl);
        if (e != hipSuccess) return e;
        if ((e = hipMemsetAsync(sl.counter, 0, 4, stream)) != hipSuccess) return e;
        const uint32_t blocks = b.n < (uint32_t)sl.blocks ? b.n : (uint32_t)sl.blocks;
        if (level >= 6) hipLaunchKernelGGL(deflate_dyn2_kernel, dim3(blocks), dim3(64), 0, stream, a, sl.scratch, sl.counter);
        else hipLaunchKernelGGL(deflate_dyn_kernel, dim3(blocks), dim3(64), 0, stream, a, sl.scratch, sl.counter);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(deflate_kernel, dim3(b.n), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace chip
