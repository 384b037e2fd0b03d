// Batched DEFLATE inflate for gfx950 (MI355X): one wavefront decodes one independent unit.
//
// Replaces, per unit, what compu reaches through sys::inflate (src/decoder/mod.rs:470) for a
// decoder built by Interface::zlib_ng(mode) (src/decoder/zlib_ng.rs:61-90): block-header parse,
// dynamic-Huffman table build, bitstream decode, LZ77 copy (RFC 1951), and the zlib / gzip
// wrappers with Adler-32 / CRC-32 verification (RFC 1950 / 1952).
//
// Data flow per wave (inflate_kernel, the one-kernel path):  block header --> canonical-Huffman tables in LDS (two-level, see below)
//   --> per block a sequence of SUPER-ROUNDS.  A super-round stages the next 64 x 320 bits of input in LDS (coalesced dword loads)
//   from the true token boundary B on; lane i decodes the token chain that starts at B + 320 i (a guess, except for lane 0), marks
//   the token boundaries it passes inside its own 320-bit segment in a bit map of the staged input, and keeps going past the
//   segment's end until it steps on a boundary marked by the owner of the segment it is in (from there on the two chains are the
//   same), at most 1536 bits further (3072 in fixed-Huffman blocks).  Tokens (literal | length, distance) go to the lane's row of
//   the wave's scratch in HBM / L2, 16 bytes per four tokens.  The chain of joins from lane 0 is the true token stream; it is found
//   in registers (pointer doubling with ds_bpermute) and executed a chunk (<= 2.5 KB of output) at a time: 64 tokens per step
//   fetched through an LDS-DMA ring, output offsets by wave prefix sum, literals and queued matches assembled in LDS and stored
//   coalesced into the unit's output range in HBM (the LZ77 window is the output itself).
// The grid is persistent: waves take units from a counter, so the token scratch is one slot per resident wave.
//
// tokens_kernel is the same decoder with the execution left out: it appends a unit's true tokens to an arena in stream order and
// leaves a record; lz77_kernel (lz77.hip) executes them in an LDS image of the unit's output.  This two-kernel pipeline is behind
// CHIP_INFLATE_PIPE=1 ("Round 4": built, parity-green, measured slower than the one-kernel path).
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "chip_internal.h"
#include "wave_checksums.h"

namespace chip {

// pointers into HBM keep their address space through the scalar round trip of rdfirst_ptr (else every access through them is a
// flat one: slower, and counted as an LDS access too)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));  // at any byte address (gfx950 takes any alignment for global and LDS accesses)
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
template <typename T>
__device__ __forceinline__ GAS T *rdfirst_gptr(T *p)
{
    return (GAS T *)rdfirst_ptr(p);
}

// phases of a unit's work: always inlined into the kernel (LDS accesses through a WaveLds reference become flat accesses in an
// out-of-line function: measured 25 % slower in round 2)
#ifndef CHIP_PHASE_FN
#define CHIP_PHASE_FN __attribute__((always_inline))
#endif

// Diagnostic build (-DCHIP_STATS): per-unit cycle and event counters; compiled out of the product.
#ifdef CHIP_STATS
#define STAT_DECL unsigned long long st_[24] = {0}; unsigned long long st_t0_ = 0
#define STAT_T0() (st_t0_ = __builtin_readcyclecounter())
#define STAT_ACC(i) do { unsigned long long n_ = __builtin_readcyclecounter(); st_[i] += n_ - st_t0_; st_t0_ = n_; } while (0)
#define STAT_ADD(i, v) (st_[i] += (unsigned long long)(v))
#define STAT_PARAM , unsigned long long *st_, unsigned long long &st_t0_
#define STAT_ARG , st_, st_t0_
#else
#define STAT_DECL
#define STAT_T0()
#define STAT_ACC(i)
#define STAT_ADD(i, v)
#define STAT_PARAM
#define STAT_ARG
#endif
// stat slots (cycles): 0 other header work, 1 walk set-up (first window chunks), 2 walk, 3 path resolve, 4 code lengths (+ block
// header), 5 table build, 6 checksum/trailer, 7 flush preparation (piece entry ranks), 16 token groups (fetch, placement,
// literals, match queue), 17 match rounds, 18 chunk store, 20 rest of the flush; (counts): 8 walk rounds (macro-rounds), 9 pieces
// on the path, 10 tokens, 11 walk trips (8 tokens each), 12 parallel copy passes, 13 chunks, 14 match rounds, 15 matches
// copied by the whole wave, 19 pieces claimed, 21 lane-tokens decoded by the walk

// ---- code-length alphabet: table entry format of round 1 (one level, 7-bit root) --------------------
// [3:0] code length, [31:16] symbol
enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_BAD = 3 };
__device__ __forceinline__ constexpr uint32_t mk_entry(uint32_t cl, uint32_t eb, uint32_t kind, uint32_t base)
{
    return cl | (eb << 4) | (kind << 8) | (base << 16);
}

constexpr int LIT_ROOT = 9;
constexpr int DIST_ROOT = 8;
constexpr int CL_ROOT = 7;

// ---- literal/length and distance tables ------------------------------------------------------------
// Literal/length codes are always looked up in two steps, without a branch (with a 9-bit root 11 % of this
// workload's tokens have longer codes, so among 64 lanes the long path ran at every step anyway):
//   lit_root[next 9 bits] (16 bits) = [4:0] number of further index bits nb, [15:5] index in pool[] of the code's
//   entry (nb = 0) or of its prefix's sub-table (2^nb entries, indexed by the nb bits behind the root's nine);
//   pool[] entry (32 bits, "final"): [4:0] code length cl, [9:5] number of extra bits eb, [14:10] cl + eb,
//   [15] halt (end of block, or with [25]: invalid code), [24:16] value (literal byte, or length base 3..258),
//   [31] length code.  The fields sit where v_bfe_u32 / v_alignbit_b32 read their 5-bit offset and width operands.
// Distance codes: dist_root[next 8 bits] (16 bits) is the final entry for codes of up to 8 bits (99 % of matches):
//   [3:0] code length, [7:4] number of extra bits, [9:8] mantissa (distance - 1 = mantissa << extra bits, plus the extra
//   bits' value), [10] invalid code, [11] longer code: then [3:0] is the number of further index bits and
//   {[15:12], [10:4]} the index (in 16-bit units) of a sub-table of such entries at the top of pool[].
constexpr uint32_t F_HALT = 1u << 15, F_INV = 1u << 25, F_LEN = 1u << 31;
constexpr uint32_t D_BAD = 1u << 10, D_LONG = 1u << 11;

__device__ __forceinline__ uint32_t make_final(uint32_t sym, uint32_t len)
{
    if (sym < 256) return len | (len << 10) | (sym << 16);
    if (sym == 256) return len | (len << 10) | F_HALT;
    if (sym >= 286) return len | (len << 10) | F_HALT | F_INV;
    const uint32_t s = sym - 257;
    uint32_t eb = 0, base = 3 + s;
    if (s == 28) base = 258;
    else if (s >= 8) {
        eb = (s >> 2) - 1;
        base = 3 + ((4 + (s & 3)) << eb);
    }
    return len | (eb << 5) | ((len + eb) << 10) | (base << 16) | F_LEN;
}
__device__ __forceinline__ uint32_t make_dist16(uint32_t sym, uint32_t len)
{
    if (sym >= 30) return len | D_BAD;
    if (sym < 4) return len | (sym << 8);
    return len | (((sym >> 1) - 1) << 4) | ((2 + (sym & 1)) << 8);
}

// ---- geometry of the speculative wave-parallel walk --------------------------------------------------
// Per super-round lane i walks the token chain that starts at B + i*S_BITS (a guess, except for lane 0), marks the token
// boundaries it passes inside its own segment in a bit map of the staged input (one mark bit per input bit), and keeps walking
// past its segment until it steps on a boundary marked by the owner of the segment it is in (from there on the two chains are
// the same), at most XT_BITS further.  Every lane records up to ROW_TOKENS tokens.  S_BITS is an odd number of dwords so that
// the 64 lanes' first window reads hit distinct LDS banks.
#ifndef CHIP_S_BITS  // geometry overridable for experiments (round 4: 320-bit segments and 2.5 KB chunks let 18 waves per CU in, 8960 B of LDS
                     // each; at the full 65 536-unit launch that is 14.6 ms against 15.1 for 384 bits / 3.5 KB at 16 waves, 15.6 for this geometry at 16,
                     // 15.5 at 17 waves with 352 bits, 15.3 at 19 with 288, 15.9 at 20 with 224 -- DESIGN.md sec. 8)
#define CHIP_S_BITS 320
#define CHIP_XT_BITS 1536
#define CHIP_XT_BITS_FIXED 3072  // fixed-Huffman codes (nearly all 8 or 9 bits) fall into step slowly
#define CHIP_ROW_TOKENS 256
#endif
constexpr uint32_t S_BITS = CHIP_S_BITS;
constexpr uint32_t XT_BITS = CHIP_XT_BITS;
#ifndef CHIP_XT_BITS_FIXED
#define CHIP_XT_BITS_FIXED CHIP_XT_BITS
#endif
constexpr uint32_t XT_BITS_FIXED = CHIP_XT_BITS_FIXED;
constexpr uint32_t ROW_TOKENS = CHIP_ROW_TOKENS;
static_assert(S_BITS % 32 == 0 && ROW_TOKENS % 4 == 0 && 64 * ROW_TOKENS < 65536, "geometry");
constexpr uint32_t SEG_WORDS = S_BITS / 32 + 1;  // mark words a segment can touch
// x / S_BITS for x < 2^15 (bit offsets inside a super-round) as a multiply and a shift
constexpr uint32_t SEG_SHIFT = 24;  // (exact for every geometry tried: 224 .. 384-bit segments; checked below)
constexpr uint32_t SEG_MAGIC = ((1u << SEG_SHIFT) + S_BITS - 1) / S_BITS;
constexpr bool seg_magic_ok()
{
    for (uint32_t x = 0; x < 64u * S_BITS + 4096u; x++)
        if (((x * SEG_MAGIC) >> SEG_SHIFT) != x / S_BITS) return false;
    return true;
}
static_assert(seg_magic_ok() && (64u * S_BITS + 4096u) * (uint64_t)SEG_MAGIC < (1ull << 32), "segment index by multiplication");
constexpr uint32_t WIN_DW = (31 + 64 * S_BITS + 48 + 96 + 31) / 32 + 3;  // staged input of a super-round, dwords
constexpr uint32_t HDR_IN_DW = 192;    // input window of the block-header parser, dwords (the code lengths take <= 4584 bits)

// Scratch of a wave in HBM: the lanes' token rows.  Layout: a 128-byte line holds four tokens (one 16-byte store) of each of eight neighbouring lanes, so that one
// store instruction of the walk fills whole lines (lane l, row token k = 4q + j -> word rowbase(l) + 32 q + j with
// rowbase(l) = (l / 8) * 8 * ROW_TOKENS + (l % 8) * 4).
constexpr size_t ROWS_WORDS = (size_t)64 * ROW_TOKENS;
__device__ __forceinline__ uint32_t row_base(uint32_t l) { return (l >> 3) * (8u * ROW_TOKENS) + (l & 7u) * 4u; }
__device__ __forceinline__ uint32_t row_word(uint32_t k) { return k + (k >> 2) * 28u; }  // 32 * (k / 4) + k % 4

// token: [8:0] literal byte, or match length 3..258; [9] match; [25:10] match distance - 1 (a piece's entry token in a joined lane's
// row is found from the popcount of the owner's boundary marks, not from the tokens)

struct HuffMeta {
    uint32_t limit15[16];  // [l] = end (exclusive) of the 15-bit-aligned code space of lengths <= l; [0] = 0
    uint32_t offs[16];     // [l] = index in sorted[] of the first symbol of length l
    uint32_t maxlen;
};

// LZ77 execution state of a chunk (see below)
#ifndef CHIP_CHUNK_BYTES
#define CHIP_CHUNK_BYTES 2560
#endif
#ifndef CHIP_COPY_LANE_MAX
#define CHIP_COPY_LANE_MAX 32
#endif
constexpr uint32_t CHUNK_BYTES = CHIP_CHUNK_BYTES;
constexpr uint32_t COPY_LANE_MAX = CHIP_COPY_LANE_MAX;
constexpr uint32_t MQ_CAP = 128;  // queued matches: at most 63 left over + 64 new
constexpr uint32_t IMG_WORDS = CHUNK_BYTES / 4 + 8;  // + room for the 16-byte reads that run past a source's end
constexpr uint32_t TOK_RING = 4;   // token groups on their way from the scratch rows into LDS (LDS-DMA: no registers held)
static_assert(CHUNK_BYTES % 4 == 0 && CHUNK_BYTES >= 1024 && COPY_LANE_MAX % 16 == 0 && IMG_WORDS % 2 == 0, "geometry");

constexpr uint32_t FLUSH_BYTES = 4 * (IMG_WORDS + 2 * MQ_CAP + 64 * TOK_RING + 128);
constexpr uint32_t HDR_BYTES = 4 * HDR_IN_DW + 4 * (1 << CL_ROOT) + 80 + sizeof(HuffMeta) + 64 + 320 + 2 * 288 + 2 * 32 + 2 * sizeof(HuffMeta);
constexpr uint32_t PHASE_BYTES = 8 * WIN_DW > FLUSH_BYTES ? (8 * WIN_DW > HDR_BYTES ? 8 * WIN_DW : HDR_BYTES) : (FLUSH_BYTES > HDR_BYTES ? FLUSH_BYTES : HDR_BYTES);
#ifndef CHIP_LDS_BYTES
#define CHIP_LDS_BYTES 8960  // per wave: 18 waves per CU (the occupancy query: 16 at 10240 B, 17 at 9600, 19 at 8448, 20 at 8192; CHIP_DEBUG_GRID prints it)
#endif
constexpr uint32_t POOL_WORDS = (CHIP_LDS_BYTES - 2 * 512 - 2 * 256 - PHASE_BYTES) / 4;
constexpr uint32_t POOL_U16 = 2 * POOL_WORDS;

struct alignas(16) WaveLds {
    union {  // first: the walk's window reads (ds_read2_b32) take small offsets only
        struct {                             // walk
            uint32_t win[WIN_DW];            // input dwords from the super-round's first on
            uint32_t bm[WIN_DW];             // mark bits, one per input bit
        } w;
        struct {                             // block header and table build
            uint32_t inbuf[HDR_IN_DW];
            uint32_t cl_lut[1 << CL_ROOT];
            uint32_t cl_sorted[20];
            HuffMeta cl_h;
            uint32_t count[16];
            uint8_t lens[320];
            uint16_t lit_sorted[288];        // symbols in canonical order
            uint16_t dist_sorted[32];
            HuffMeta lit_h, dist_h;
        } hdr;
        struct {                             // LZ77 execution
            uint32_t out[IMG_WORDS];         // the chunk's output bytes by offset (offset 0 = the dword-aligned address below the chunk)
            uint2 mq[MQ_CAP];                // queued match: .x offset of its first output byte, .y length | distance << 16
            uint32_t tok[64 * TOK_RING];     // token groups, written by global_load_lds_dword
            uint32_t pk[128];                // k-th piece of a batch: [2k] its first stream index, [2k+1] [15:0] (row token - stream index) mod 2^16, [31:16] row base
        } fl;
        uint32_t phase_words_[PHASE_BYTES / 4];
    };
    uint16_t lit_root[1 << LIT_ROOT];
    uint16_t dist_root[1 << DIST_ROOT];
    uint32_t pool[POOL_WORDS];  // literal/length finals and sub-tables from the bottom, distance sub-tables (16-bit) from the top
};
static_assert(sizeof(WaveLds) <= CHIP_LDS_BYTES, "the waves per CU that CHIP_LDS_BYTES stands for");
// pool[] cannot overflow.  In a canonical code the codes of one length are neighbours, so every 9-bit prefix that lies inside the codes
// of length L > 9 has a sub-table of 2^(L-9) entries, one per code; only the (at most one per length) prefixes that straddle two lengths
// hold entries that repeat a code.  Literal/length: <= 286 codes + 1 invalid entry + (2 + 4 + ... + 64) = 413 words; distance (16-bit
// entries, 8-bit root): <= 30 + (2 + 4 + ... + 128) = 284 entries = 142 words.
static_assert(POOL_WORDS >= 413 + 142, "the tables of any valid block fit");
// the CRC-32 tables (2048 words) take the whole structure: nothing else is live while a checksum runs
#ifndef CHIP_EXP_ALLOW_SMALL_LDS  // (timing probes of raw-deflate batches only: gzip units would run over the structure)
static_assert(sizeof(WaveLds) >= 2048 * 4, "wave_crc32 needs 8 KB");
#endif

// Canonical lookup: x15 = next 15 stream bits, first bit in bit 14 (MSB-first code value).  Returns the symbol's index in
// sorted[] (0xffffffff: no such code) and its length.
__device__ __forceinline__ uint32_t canon_index(const HuffMeta &H, uint32_t x15, uint32_t &l)
{
    l = 1;
    if (x15 >= H.limit15[15]) return 0xffffffffu;
#pragma unroll
    for (int j = 1; j < 15; j++) l += (x15 >= H.limit15[j]) ? 1u : 0u;
    return H.offs[l] + ((x15 - H.limit15[l - 1]) >> (15 - l));
}

// The same with the limits in scalar registers (read once per table build) and the code's length known to lie in [LO, HI]: the root
// fill knows x15 < limit15[ROOT] (lengths up to ROOT), a sub-table knows x15 >= limit15[ROOT] (lengths above it), so eight or five
// comparisons against registers stand where fourteen LDS reads and comparisons stood.
struct CanonLim {
    uint32_t lim[16];
};
__device__ __forceinline__ CanonLim canon_limits(const HuffMeta &H)
{
    CanonLim c;
#pragma unroll
    for (int j = 0; j < 16; j++) c.lim[j] = rdfirst(H.limit15[j]);
    return c;
}
template <int LO, int HI>
__device__ __forceinline__ uint32_t canon_index_in(const HuffMeta &H, const CanonLim &c, uint32_t x15, uint32_t &l)
{
    l = LO;
#pragma unroll
    for (int j = LO; j < HI; j++) l += (x15 >= c.lim[j]) ? 1u : 0u;
    return H.offs[l] + ((x15 - H.limit15[l - 1]) >> (15 - l));
}

// Code lengths -> canonical description H and the symbols in canonical order (RFC 1951 sec. 3.2.2), with zlib's acceptance
// rules.  Wave-cooperative; lens[], count[] live in LDS.  Returns 0, -1 (invalid set) or 1 (empty set); uniform.
template <typename SortedT, typename MakeT>
__device__ __forceinline__ int canon_prep(uint32_t *count, const uint8_t *lens, int n_, bool code_lengths, SortedT *sorted, HuffMeta &H, MakeT make)
{
    const uint32_t lane = lane_id();
    const int n = (int)rdfirst((uint32_t)n_);
    if (lane < 16) count[lane] = 0;
    WSYNC();
    for (int s = lane; s < n; s += 64) {
        uint32_t l = lens[s];
        if (l) atomicAdd(&count[l], 1u);
    }
    WSYNC();
    // Lane l (1..15) holds the count of length l.  offs[l] = symbols of shorter lengths; limit15[l] = sum over j <= l of
    // count[j] << (15 - j): the end of the code space that lengths <= l take, aligned to 15 bits.  zlib's running `left` is
    // (2^15 - limit15[l]) >> (15 - l): the set is over-subscribed when some limit15[l] exceeds 2^15 and incomplete when
    // limit15[15] stays below it.  (Two wave prefix sums instead of a 15-step chain of LDS reads and scalar updates.)
    const uint32_t c = lane >= 1 && lane <= 15 ? count[lane] : 0u;
    const uint32_t cincl = wave_incl_scan(c);
    const uint32_t off = cincl - c;
    const uint32_t lim = wave_incl_scan(lane <= 15 ? c << (15u - (lane & 15u)) : 0u);
    const uint32_t present = (uint32_t)__ballot(c != 0);
    const uint32_t maxlen = present ? 31u - (uint32_t)__clz((int)present) : 0u;
    const bool over = __any(lane <= 15 && lim > 32768u);
    const int left = rdlane(lim, 15) < 32768u ? 1 : 0;
    uint32_t mynext = off;
    if (lane <= 15) {
        H.limit15[lane] = lim;
        H.offs[lane] = off;
    }
    if (lane == 0) H.maxlen = maxlen;
    WSYNC();
    if (maxlen == 0) return 1;
    if (over) return -1;
    if (left > 0 && (code_lengths || maxlen != 1)) return -1;
    // stable counting sort by (length, symbol)
    for (int base = 0; base < n; base += 64) {
        int s = base + (int)lane;
        uint32_t l = s < n ? lens[s] : 0;
        for (uint32_t pm = present & 0xfffeu; pm; pm &= pm - 1u) {  // the lengths that occur
            const uint32_t ll = (uint32_t)__builtin_ctz(pm);
            uint64_t m = __ballot(l == ll);
            uint32_t bp = rdlane(mynext, ll);
            if (l == ll) sorted[bp + __popcll(m & lanemask_lt())] = make((uint32_t)s, l);
            if (lane == ll) mynext += __popcll(m);
        }
    }
    WSYNC();
    return 0;
}

// The code-length alphabet's one-level table (zlib: a set made only of zeros reads as symbol 0 of length 1).
__device__ CHIP_PHASE_FN int build_cl_table(WaveLds &L)
{
    const uint32_t lane = lane_id();
    HuffMeta &H = L.hdr.cl_h;
    const int r = canon_prep(L.hdr.count, L.hdr.lens, 19, true, L.hdr.cl_sorted, H, [](uint32_t s, uint32_t l) { return mk_entry(l, 0, K_LIT, s); });
    if (r < 0) return -1;
    for (uint32_t idx = lane; idx < (1u << CL_ROOT); idx += 64) {
        uint32_t e = mk_entry(1, 0, K_LIT, 0);
        if (r == 0) {
            uint32_t l;
            const uint32_t ci = canon_index(H, __brev(idx) >> 17, l);  // codes are at most 7 bits: the root covers them all
            e = ci == 0xffffffffu ? mk_entry(H.maxlen, 0, K_BAD, 0) : L.hdr.cl_sorted[ci];
        }
        L.hdr.cl_lut[idx] = e;
    }
    WSYNC();
    return 0;
}

// Literal/length tables of a block from lens[0..n).  Returns 0, -1 (invalid set), or 1 (the codes' entries do not fit pool[]: cannot
// happen, see POOL_WORDS).  `used` = words of pool[] taken.
__device__ CHIP_PHASE_FN int build_litlen(WaveLds &L, const uint8_t *lens, int n, uint32_t &used)
{
    const uint32_t lane = lane_id();
    HuffMeta &H = L.hdr.lit_h;
    uint16_t *const sorted = L.hdr.lit_sorted;
    const int r = canon_prep(L.hdr.count, lens, n, false, sorted, H, [](uint32_t s, uint32_t) { return (uint16_t)s; });
    if (r != 0) return -1;  // no literal/length code at all cannot be: the end-of-block code exists
    const CanonLim CL = canon_limits(H);
    const uint32_t lim_r = CL.lim[LIT_ROOT], top = CL.lim[15];
    const uint32_t nshort = rdfirst(H.offs[LIT_ROOT + 1 <= 15 ? LIT_ROOT + 1 : 15]);  // symbols with codes of up to nine bits come first
    const uint32_t nsym = rdfirst(H.offs[15]) + rdfirst(L.hdr.count[15]);
    const uint32_t nfirst = rdfirst(H.maxlen) <= (uint32_t)LIT_ROOT ? nsym : nshort;
    if (nfirst + 1 > POOL_WORDS) return 1;
    for (uint32_t i = lane; i < nfirst; i += 64) {
        const uint32_t s = sorted[i];
        L.pool[i] = make_final(s, lens[s]);
    }
    const uint32_t bad_idx = nfirst;
    const uint32_t bad_e = rdfirst(H.maxlen) | (rdfirst(H.maxlen) << 10) | F_HALT | F_INV;
    if (lane == 0) L.pool[bad_idx] = bad_e;
    for (uint32_t idx = lane; idx < (1u << LIT_ROOT); idx += 64) {
        const uint32_t x15 = __brev(idx) >> 17;
        if (x15 >= top) L.lit_root[idx] = (uint16_t)(bad_idx << 5);
        else if (x15 < lim_r) {
            uint32_t l;
            L.lit_root[idx] = (uint16_t)(canon_index_in<1, LIT_ROOT>(H, CL, x15, l) << 5);
        }
    }
    used = nfirst + 1;
    if (top > lim_r) {
        // sub-tables: one per root prefix that holds longer codes; the longest code of a prefix is its last one
        constexpr uint32_t P = 1u << (15 - LIT_ROOT);
        const uint32_t np = (top - lim_r + P - 1) / P;
        for (uint32_t j0 = 0; j0 < np; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool have = j < np;
            const uint32_t xj = lim_r + j * P;
            uint32_t sb = 0;
            if (have) {
                uint32_t l;
                (void)canon_index_in<LIT_ROOT + 1, 15>(H, CL, xj + P < top ? xj + P - 1 : top - 1, l);
                sb = l - LIT_ROOT;
            }
            const uint32_t size = have ? (1u << sb) : 0u;
            const uint32_t incl = wave_incl_scan(size);
            const uint32_t off = used + incl - size;
            const uint32_t total = rdlane(incl, 63);
            if (used + total > POOL_WORDS) return 1;
            if (have) {
                for (uint32_t t = 0; t < size; t++) {
                    const uint32_t x15 = sb ? xj + ((__brev(t) >> (32 - sb)) << (15 - LIT_ROOT - sb)) : xj;
                    uint32_t l, e = bad_e;
                    if (x15 < top) {
                        const uint32_t s = sorted[canon_index_in<LIT_ROOT + 1, 15>(H, CL, x15, l)];
                        e = make_final(s, l);
                    }
                    L.pool[off + t] = e;
                }
                L.lit_root[__brev(xj >> (15 - LIT_ROOT)) >> (32 - LIT_ROOT)] = (uint16_t)((off << 5) | sb);
            }
            used += total;
        }
    }
    WSYNC();
    return 0;
}

// Distance tables from lens[0..n); `used` = pool words the literal/length tables took.  Returns 0, -1 or 1 as build_litlen.
__device__ CHIP_PHASE_FN int build_dist(WaveLds &L, const uint8_t *lens, int n, uint32_t used)
{
    const uint32_t lane = lane_id();
    HuffMeta &H = L.hdr.dist_h;
    uint16_t *const sorted = L.hdr.dist_sorted;
    const int r = canon_prep(L.hdr.count, lens, n, false, sorted, H, [](uint32_t s, uint32_t) { return (uint16_t)s; });
    if (r < 0) return -1;
    if (r == 1) {  // no distance code: zlib accepts the block, every match is invalid
        for (uint32_t idx = lane; idx < (1u << DIST_ROOT); idx += 64) L.dist_root[idx] = (uint16_t)(1u | D_BAD);
        WSYNC();
        return 0;
    }
    const CanonLim CL = canon_limits(H);
    const uint32_t lim_r = CL.lim[DIST_ROOT], top = CL.lim[15], maxlen = rdfirst(H.maxlen);
    uint16_t *const pool16 = (uint16_t *)L.pool;
    for (uint32_t idx = lane; idx < (1u << DIST_ROOT); idx += 64) {
        const uint32_t x15 = __brev(idx) >> 17;
        if (x15 >= top) L.dist_root[idx] = (uint16_t)(maxlen | D_BAD);
        else if (x15 < lim_r) {
            uint32_t l;
            const uint32_t s = sorted[canon_index_in<1, DIST_ROOT>(H, CL, x15, l)];
            L.dist_root[idx] = (uint16_t)make_dist16(s, l);
        }
    }
    if (top > lim_r) {
        constexpr uint32_t P = 1u << (15 - DIST_ROOT);
        const uint32_t np = (top - lim_r + P - 1) / P;  // at most 15: thirty symbols, two per longer prefix
        uint32_t used16 = 0;
        for (uint32_t j0 = 0; j0 < np; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool have = j < np;
            const uint32_t xj = lim_r + j * P;
            uint32_t sb = 0;
            if (have) {
                uint32_t l;
                (void)canon_index_in<DIST_ROOT + 1, 15>(H, CL, xj + P < top ? xj + P - 1 : top - 1, l);
                sb = l - DIST_ROOT;
            }
            const uint32_t size = have ? (1u << sb) : 0u;
            const uint32_t incl = wave_incl_scan(size);
            const uint32_t total = rdlane(incl, 63);
            if (2 * used + used16 + total > POOL_U16) return 1;
            const uint32_t base16 = POOL_U16 - used16 - incl;  // sub-tables grow downwards from the top
            if (have) {
                for (uint32_t t = 0; t < size; t++) {
                    const uint32_t x15 = sb ? xj + ((__brev(t) >> (32 - sb)) << (15 - DIST_ROOT - sb)) : xj;
                    uint32_t l, e = maxlen | D_BAD;
                    if (x15 < top) {
                        const uint32_t s = sorted[canon_index_in<DIST_ROOT + 1, 15>(H, CL, x15, l)];
                        e = make_dist16(s, l);
                    }
                    pool16[base16 + t] = (uint16_t)e;
                }
                L.dist_root[__brev(xj >> (15 - DIST_ROOT)) >> (32 - DIST_ROOT)] = (uint16_t)(D_LONG | sb | ((base16 & 127u) << 4) | ((base16 >> 7) << 12));
            }
            used16 += total;
        }
    }
    WSYNC();
    return 0;
}
struct InWin {
    const uint32_t *g32;  // dword-aligned base covering the unit's bytes
    uint32_t total_dw;    // dwords that contain at least one byte of the unit
    uint32_t win0;        // header parser: dword index held in hdr.inbuf[0]
};


__device__ __forceinline__ void win_load(WaveLds &L, InWin &w, uint32_t dw_start)
{
    WSYNC();
    w.win0 = dw_start;
    // all loads of the window go out before the first LDS store waits for one
    constexpr uint32_t PER_LANE = (HDR_IN_DW + 63) / 64;
    uint32_t v[PER_LANE];
#pragma unroll
    for (uint32_t j = 0; j < PER_LANE; j++) {
        const uint32_t i = dw_start + 64u * j + lane_id();
        v[j] = i < w.total_dw ? w.g32[i] : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < PER_LANE; j++) {
        const uint32_t k = 64u * j + lane_id();
        if (k < HDR_IN_DW) L.hdr.inbuf[k] = v[j];
    }
    WSYNC();
}

// make sure the dwords covering `span` bits from `pos` (plus one for the funnel shift) are staged
__device__ __forceinline__ void win_ensure(WaveLds &L, InWin &w, uint32_t pos, uint32_t span = 64)
{
    uint32_t d = pos >> 5;
    if (d < w.win0 || d + ((span + 62u) >> 5) + 1u > w.win0 + HDR_IN_DW) win_load(L, w, d);
}

// 64 stream bits starting at `pos` (lo = first 32)
__device__ __forceinline__ void win_bits(const WaveLds &L, const InWin &w, uint32_t pos, uint32_t &lo, uint32_t &hi)
{
    uint32_t D = (pos >> 5) - w.win0, sh = pos & 31;
    uint32_t d0 = L.hdr.inbuf[D], d1 = L.hdr.inbuf[D + 1], d2 = L.hdr.inbuf[D + 2];
    lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    hi = __builtin_amdgcn_alignbit(d2, d1, sh);
}

// the same at a wave-uniform position: the result is the same in every lane, and is handed back as such (scalar
// registers; branches on header fields become scalar branches instead of exec-mask regions)
__device__ __forceinline__ void win_bits_uniform(const WaveLds &L, const InWin &w, uint32_t pos, uint32_t &lo, uint32_t &hi)
{
    uint32_t l, h;
    win_bits(L, w, pos, l, h);
    lo = rdfirst(l);
    hi = rdfirst(h);
}

__device__ __forceinline__ uint32_t bfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }

// zlib / gzip header (RFC 1950 sec. 2.2, RFC 1952 sec. 2.3), checks in zlib's order.  Returns
// ST_RUNNING and the header length in bytes, or the final status (need-input / error / need-dict).
__device__ int32_t parse_wrapper(LDS_AS uint32_t *tab, const uint8_t *gin, uint32_t avail, int32_t format, uint32_t &wrap, uint32_t &hdr)
{
    wrap = 0;
    hdr = 0;
    if (avail < 2) return CHIP_NEED_INPUT;
    const uint32_t b0 = gin[0], b1 = gin[1];
    const bool allow_gzip = format == CHIP_FMT_GZIP || format == CHIP_FMT_AUTO;
    const bool allow_zlib = format == CHIP_FMT_ZLIB || format == CHIP_FMT_AUTO;
    if (allow_gzip && b0 == 0x1f && b1 == 0x8b) {
        wrap = 2;
        if (avail < 4) return CHIP_NEED_INPUT;
        const uint32_t flg = gin[3];
        if (gin[2] != 8) return Z_DATA_ERROR;  // unknown compression method
        if (flg & 0xe0) return Z_DATA_ERROR;   // unknown header flags set
        if (avail < 10) return CHIP_NEED_INPUT;
        uint32_t k = 10;
        if (flg & 4) {  // FEXTRA
            if (avail < k + 2) return CHIP_NEED_INPUT;
            uint32_t xlen = gin[k] | ((uint32_t)gin[k + 1] << 8);
            k += 2;
            if (avail - k < xlen) return CHIP_NEED_INPUT;
            k += xlen;
        }
        for (uint32_t bit = 8; bit <= 16; bit <<= 1) {  // FNAME, FCOMMENT: zero-terminated
            if (!(flg & bit)) continue;
            for (;;) {
                uint32_t idx = k + lane_id();
                uint64_t z = __ballot(idx < avail && gin[idx] == 0);
                if (z) {
                    k += (uint32_t)__ffsll((long long)z);
                    break;
                }
                if (avail - k <= 64) return CHIP_NEED_INPUT;
                k += 64;
            }
        }
        if (flg & 2) {  // FHCRC: low 16 bits of the CRC-32 of the header so far
            if (avail < k + 2) return CHIP_NEED_INPUT;
            uint32_t crc = wave_crc32(tab, gin, k);
            if ((crc & 0xffffu) != (gin[k] | ((uint32_t)gin[k + 1] << 8))) return Z_DATA_ERROR;
            k += 2;
        }
        hdr = k;
        return ST_RUNNING;
    }
    if (!allow_zlib) return Z_DATA_ERROR;                  // incorrect header check
    if (((b0 << 8) | b1) % 31u) return Z_DATA_ERROR;       // incorrect header check
    if ((b0 & 15u) != 8) return Z_DATA_ERROR;              // unknown compression method
    if ((b0 >> 4) + 8 > 15) return Z_DATA_ERROR;           // invalid window size
    wrap = 1;
    hdr = 2;
    if (b1 & 0x20) {  // FDICT: compu never sets a dictionary, zlib answers Z_NEED_DICT after the id
        if (avail < 6) return CHIP_NEED_INPUT;
        hdr = 6;
        return CHIP_NEED_DICT;
    }
    return ST_RUNNING;
}

// unaligned accesses (gfx950 handles any byte alignment for LDS and global dword accesses); LDS pointers carry their
// address space so that a choice between an LDS and a global source never turns into a flat access
struct __attribute__((packed)) U32u { uint32_t v; };
struct __attribute__((packed)) U16u { uint16_t v; };
struct __attribute__((packed)) U128u { uint32_t x, y, z, w; };
typedef LDS_AS uint8_t lds_u8;

// Stored-block payload: copy n bytes from byte offset `so` of the unit's input to gdst.  Byte copies bring the
// destination to 16-byte alignment, then every lane moves 16 bytes per piece, two pieces in flight: one (unaligned) dwordx4
// load straight from the source -- gfx950 takes any byte alignment -- and one aligned dwordx4 store: a memcpy at HBM rate.
__device__ void wave_copy_stored(uint8_t *gdst, const uint32_t *g32, uint32_t total_dw, uint32_t so, uint32_t n)
{
    const uint32_t lane = lane_id();
    const uint8_t *gsrc = (const uint8_t *)g32 + so;
    uint32_t head = (uint32_t)((16u - ((uintptr_t)gdst & 15u)) & 15u);
    if (head > n) head = n;
    if (lane < head) gdst[lane] = gsrc[lane];
    uint32_t done = head;
    const uint32_t body = (n - done) & ~15u;  // every 16-byte read lies inside the n source bytes
    if (body) {
        uint4 *d16 = (uint4 *)(gdst + done);
        const U128u *s16 = (const U128u *)(gsrc + done);
        const uint32_t cnt = body >> 4;
        uint32_t i = lane;
#ifdef CHIP_STORED_NT_LD
#define CHIP_STORED_LD(p) __builtin_nontemporal_load(p)
#else
#define CHIP_STORED_LD(p) (*(p))
#endif
#ifndef CHIP_STORED_PLAIN_ST  // stored bytes are written once: the non-temporal hint keeps them from pushing other units' windows out of L2 (0.507 -> 0.476 ms per 16 384 units)
#define CHIP_STORED_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define CHIP_STORED_ST(v, p) (*(p) = (v))
#endif
#ifndef CHIP_STORED_DEPTH
#define CHIP_STORED_DEPTH 2  // 16-byte pieces per lane in flight (4 and 8 measured slower: 1.62 / 1.65 / 1.70 ms per 65 536 units)
#endif
        for (; i + 64u * (CHIP_STORED_DEPTH - 1) < cnt; i += 64u * CHIP_STORED_DEPTH) {
            u32x4 v[CHIP_STORED_DEPTH];
#pragma unroll
            for (int k = 0; k < CHIP_STORED_DEPTH; k++) v[k] = CHIP_STORED_LD((const u32x4_u *)(s16 + i + 64u * k));
#pragma unroll
            for (int k = 0; k < CHIP_STORED_DEPTH; k++) CHIP_STORED_ST(v[k], (u32x4 *)(d16 + i + 64u * k));
        }
        for (; i < cnt; i += 64u) {
            const U128u a = s16[i];
            d16[i] = make_uint4(a.x, a.y, a.z, a.w);
        }
        done += body;
    }
    const uint32_t tail = n - done;
    if (lane < tail) gdst[done + lane] = gsrc[done + lane];
    (void)total_dw;
}

// (off % d) for off, d < 512 without an integer divide
__device__ __forceinline__ uint32_t small_mod(uint32_t off, uint32_t d)
{
    uint32_t q = (uint32_t)(((float)off + 0.5f) * __builtin_amdgcn_rcpf((float)d));
    return off - q * d;
}
// ---- LZ77 execution -----------------------------------------------------------------------------
// The true token stream of a walk round is executed a chunk at a time.  A chunk is as many tokens as
// produce at most CHUNK_BYTES of output; the chunk's output is assembled in LDS (the image) and leaves
// with coalesced dword stores.  Tokens are taken 64 at a time (lane = token): a wave prefix sum of the
// output lengths places every token, literal lanes drop their byte into the image at once, match lanes
// append {position, length, distance} to a queue in LDS.  Whenever 64 matches are queued they are
// executed one per lane: a lane whose source is complete (entirely below the chunk: the unit's earlier
// output in HBM; or inside the image below the first match of the round) copies up to COPY_LANE_MAX
// bytes with 16-byte unaligned loads and exact-length unaligned LDS stores; what is left (sources that
// depend on matches of the same round, self-overlapping, very long or chunk-straddling matches) is done
// in further passes or, when few, one match at a time by the whole wave.
struct ChunkLds {
    uint32_t *out;  // [IMG_WORDS] the chunk's output bytes by offset (offset 0 = the dword-aligned address below the chunk)
    uint2 *mq;      // [MQ_CAP] queued match: .x offset of its first output byte, .y length | distance << 16
    uint32_t *tok;  // [64 * TOK_RING] token groups, written by global_load_lds_dword
};

__device__ __forceinline__ ChunkLds chunk_lds(WaveLds &L)
{
    ChunkLds c;
    c.out = L.fl.out;
    c.mq = L.fl.mq;
    c.tok = L.fl.tok;
    return c;
}

// exactly n (1..16) bytes of v to the LDS address d
__device__ __forceinline__ void lds_put(lds_u8 *d, const U128u &v, uint32_t n)
{
    if (n >= 4) ((LDS_AS U32u *)d)->v = v.x;
    if (n >= 8) ((LDS_AS U32u *)(d + 4))->v = v.y;
    if (n >= 12) ((LDS_AS U32u *)(d + 8))->v = v.z;
    if (n >= 16) ((LDS_AS U32u *)(d + 12))->v = v.w;
    uint32_t w = n < 4 ? v.x : n < 8 ? v.y : n < 12 ? v.z : v.w;
    lds_u8 *t = d + (n & 12u);
    if (n < 16 && (n & 2u)) {
        ((LDS_AS U16u *)t)->v = (uint16_t)w;
        w >>= 16;
        t += 2;
    }
    if (n < 16 && (n & 1u)) *t = (uint8_t)w;
}

// One queued match, copied by the whole wave (any distance, any length, source anywhere below it).
__device__ __forceinline__ void copy_one_match(lds_u8 *img, const uint8_t *base, uint32_t mis, uint32_t x, uint32_t len, uint32_t dist)
{
    const uint32_t lane = lane_id();
    const int32_t sx = (int32_t)x - (int32_t)dist;
    for (uint32_t i = lane; i < len; i += 64) {
        // a self-overlapping match repeats its first `dist` bytes; all of them lie below x and are complete
        const int32_t s = sx + (int32_t)(dist >= len ? i : small_mod(i, dist));
        uint8_t b = 0;
        if (s < (int32_t)mis) b = ((GAS const uint8_t *)base)[s];
        if (s >= (int32_t)mis) b = img[s];
        img[x + i] = b;
    }
    LSYNC();
}

// A round = up to 64 queued matches, one per lane.  round_issue() takes them off the queue and starts the loads of the
// sources that lie entirely below the chunk (final bytes in HBM: nothing in the chunk can change them); round_finish()
// copies.  The caller puts the next token group's work between the two, so the loads' latency is covered.
struct Round {
    uint32_t x, len, dist;  // first output offset, length (0 = no match in this lane), distance
    bool glob;              // source entirely below the chunk: v0 / v1 hold its first 32 bytes
    U128u v0, v1;
};

// glob_ok: 16-byte reads that start below the chunk may run up to 15 bytes into the chunk's (not yet written) output
// range without leaving the unit's capacity.
__device__ __forceinline__ void round_issue(Round &r, const ChunkLds &C, const uint8_t *base, uint32_t mis, uint32_t qh, uint32_t nr,
                                            bool glob_ok)
{
    const uint32_t lane = lane_id();
    const bool act = lane < nr;
    const uint32_t qi = (qh + lane) & (MQ_CAP - 1u);
    const uint2 e = C.mq[qi];
    r.x = e.x;
    const uint32_t ld = act ? e.y : 0u;
    r.len = ld & 0xffffu;
    r.dist = ld >> 16;
    const int32_t sx = (int32_t)r.x - (int32_t)r.dist;
    r.glob = glob_ok && r.len != 0 && r.len <= COPY_LANE_MAX && sx + (int32_t)r.len <= (int32_t)mis;
    r.v0 = U128u{0, 0, 0, 0};
    r.v1 = U128u{0, 0, 0, 0};
#ifndef CHIP_EXP_NO_GLOB  // ablation (wrong output): the LZ77 source fetches from HBM are left out
    if (r.glob) {
        const u32x4_u t = *(GAS const u32x4_u *)(base + sx);
        r.v0 = U128u{t.x, t.y, t.z, t.w};
    }
#endif
#ifndef CHIP_EXP_NO_GLOB
    if (r.glob && r.len > 16u) {
        const u32x4_u t = *(GAS const u32x4_u *)(base + sx + 16);
        r.v1 = U128u{t.x, t.y, t.z, t.w};
    }
#endif
}

__device__ __forceinline__ void round_finish(const Round &r, const ChunkLds &C, const uint8_t *base, uint32_t mis STAT_PARAM)
{
    const uint32_t lane = lane_id();
    lds_u8 *const img = (lds_u8 *)C.out;
    const int32_t sx = (int32_t)r.x - (int32_t)r.dist;
    uint64_t pend = __ballot(r.len != 0);
    STAT_ADD(14, 1);
    while (pend) {
        const uint32_t f = (uint32_t)__ffsll((long long)pend) - 1u;
        const uint32_t wp = rdlane(r.x, f);  // every byte below the first pending match is final
        const bool mine = (pend >> lane) & 1ull;
        const bool loc = sx >= (int32_t)mis && (uint32_t)sx + r.len <= wp;
        const bool ready = mine && r.len <= COPY_LANE_MAX && (r.glob || loc);
        const uint64_t rm = __ballot(ready);
        if (!((rm >> f) & 1ull)) {  // long, self-overlapping or chunk-straddling: by the whole wave
            STAT_ADD(15, 1);
            copy_one_match(img, base, mis, wp, rdlane(r.len, f), rdlane(r.dist, f));
            pend &= pend - 1ull;
            continue;
        }
        STAT_ADD(12, 1);
        uint32_t rem = ready ? r.len : 0u;
#pragma unroll
        for (uint32_t it = 0; it < COPY_LANE_MAX / 16; it++) {
            if (it && !__any(rem != 0)) break;
            U128u v = it ? r.v1 : r.v0;
            if (rem && !r.glob) {
                const lds_u8 *sp = img + sx + (int32_t)(16u * it);
                v.x = ((const LDS_AS U32u *)sp)->v;
                v.y = ((const LDS_AS U32u *)(sp + 4))->v;
                v.z = ((const LDS_AS U32u *)(sp + 8))->v;
                v.w = ((const LDS_AS U32u *)(sp + 12))->v;
            }
            if (rem) {
                const uint32_t n = rem < 16u ? rem : 16u;
                lds_put(img + r.x + 16u * it, v, n);
                rem -= n;
            }
        }
        LSYNC();
        pend &= ~rm;
    }
}

// lane i's value of x from lane `src` (any lane; ds_bpermute: no LDS memory is touched)
__device__ __forceinline__ uint32_t lane_gather(uint32_t x, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)x);
}

// ---- CHIP_F_COMPU_STATUS only (rare path): where zlib's inflate() stands when the output has filled ---------------
// zlib leaves with every bit of the token consumed during which (or, when the output was exactly full, in front of which) it ran
// out of room, and holds fewer than eight unused bits: its avail_in follows from the bit position behind that token.
// End of the token that starts at bit p (never an end-of-block code), read from memory with the block's tables.
__device__ uint32_t token_end(const WaveLds &L, const InWin &w, uint32_t p)
{
    const uint16_t *const pool16 = (const uint16_t *)L.pool;
    const uint32_t i = p >> 5;
    const uint32_t d0 = i < w.total_dw ? w.g32[i] : 0u, d1 = i + 1u < w.total_dw ? w.g32[i + 1u] : 0u, d2 = i + 2u < w.total_dw ? w.g32[i + 2u] : 0u;
    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, p), hi = __builtin_amdgcn_alignbit(d2, d1, p);
    const uint32_t r = L.lit_root[lo & ((1u << LIT_ROOT) - 1u)];
    const uint32_t e = L.pool[(r >> 5) + __builtin_amdgcn_ubfe(lo, LIT_ROOT, r)];
    const uint32_t n1 = __builtin_amdgcn_ubfe(e, 10, 5);
    if (!(e & F_LEN)) return p + n1;
    const uint32_t w2 = __builtin_amdgcn_alignbit(hi, lo, n1);
    uint32_t m = L.dist_root[w2 & ((1u << DIST_ROOT) - 1u)];
    if (m & D_LONG) {
        const uint32_t b16 = ((m >> 4) & 127u) | ((m >> 12) << 7);
        m = pool16[b16 + __builtin_amdgcn_ubfe(w2, DIST_ROOT, m)];
    }
    return p + n1 + (m & 15u) + __builtin_amdgcn_ubfe(m, 4, 4);
}

// The chunk that began with stream token `cstart` at image offset `run0` overflowed the room `xcap` (image offsets): bit position
// behind the first token that does not fit entirely.  B = the super-round's first bit (lane l's chain starts at B + l * S_BITS).
__device__ __attribute__((always_inline)) uint32_t overflow_bit(const WaveLds &L, const InWin &w, const uint32_t *grow, uint32_t ntok, uint32_t npieces,
                                                           uint32_t cstart, uint32_t run0, uint32_t xcap, uint32_t B)
{
    const uint32_t lane = lane_id();
    const uint32_t pfirst = lane < npieces ? L.fl.pk[2u * lane] : 0xffffffffu;
    const uint32_t pdelta = lane < npieces ? L.fl.pk[2u * lane + 1u] : 0u;
    uint32_t run = run0;
    for (uint32_t g = cstart; g < ntok; g += 64u) {
        const uint32_t t = g + lane < ntok ? g + lane : ntok - 1u;
        uint32_t k = 0;  // pieces that start at or before token t
        for (uint32_t q = 0; q < npieces; q++) k += rdlane(pfirst, q) <= t ? 1u : 0u;
        const uint32_t d = lane_gather(pdelta, k - 1u);
        const uint32_t krow = (t + d) & 0xffffu, rb = d >> 16;
        const uint32_t tok = grow[rb + row_word(krow)];
        uint32_t olen = (tok & 512u) ? (tok & 0x1ffu) : 1u;
        if (g + lane >= ntok) olen = 0;
        const uint32_t incl = wave_incl_scan(olen);
        const uint64_t over = __ballot(olen != 0 && run + incl > xcap);
        if (over) {
            const uint32_t f = (uint32_t)__ffsll((long long)over) - 1u;
            const uint32_t rbf = rdlane(rb, f), kf = rdlane(krow, f);
            const uint32_t owner = ((rbf / (8u * ROW_TOKENS)) << 3) | ((rbf % (8u * ROW_TOKENS)) >> 2);  // row_base() inverted
            uint32_t p = B + owner * S_BITS;
            for (uint32_t j = 0; j <= kf; j++) p = rdfirst(token_end(L, w, p));
            return p;
        }
        run += rdlane(incl, 63u);
    }
    return 0;  // (not reached: the caller saw the overflow)
}

// Executes the ntok tokens of a batch of the true stream (npieces pieces described by L.fl.pk) into gout at opos.
// Returns false when decoding must stop (error / output full).
// (ovf: on CHIP_NEED_OUTPUT the overflowing chunk's first stream token, first image offset and room, for overflow_bit())
__device__ CHIP_PHASE_FN bool flush_tokens(WaveLds &L, const uint32_t *grow_, uint32_t ntok_, uint32_t npieces_, uint8_t *gout_, uint32_t &opos_,
                             uint32_t cap_, int32_t &status, uint32_t (&ovf)[3] STAT_PARAM)
{
    const uint32_t lane = lane_id();
    // every argument is wave-uniform; say so (scalar registers, scalar loop branches)
    const uint32_t *const grow = rdfirst_ptr(grow_);
    uint8_t *const gout = rdfirst_ptr(gout_);
    const uint32_t ntok = rdfirst(ntok_), npieces = rdfirst(npieces_), cap = rdfirst(cap_);
    uint32_t opos = rdfirst(opos_);
    const ChunkLds C = chunk_lds(L);
    lds_u8 *const img = (lds_u8 *)C.out;
    if (ntok == 0) return true;
    // piece `lane`: first stream index, and what to add to a stream index to get the token's word in the scratch rows
    const uint32_t pfirst = lane < npieces ? L.fl.pk[2u * lane] : 0xffffffffu;
    const uint32_t pdelta = lane < npieces ? L.fl.pk[2u * lane + 1u] : 0u;
    // Token g + lane of the stream (lanes behind the end re-read the last token: no branch around the load).  The piece a
    // token lies in = pieces that start at or before it, minus one: `before` pieces start in front of the group (kept
    // up to date by the caller), those inside it are found through a 64-bit mask of their start positions, put together
    // with scalar instructions; a DPP-free count of the mask bits below each lane and one cross-lane read finish it.
    uint32_t before = 0;  // pieces that start before stream index gnext
    uint32_t gnext = 0;   // stream index the next fetch() asks for
    uint32_t slot_w = 0;  // ring slot the next fetch() fills
    const uint32_t tok_lds = rdfirst((uint32_t)(uintptr_t)(LDS_AS uint32_t *)C.tok);  // LDS byte address of the ring
    // The load goes straight into LDS (global_load_lds_dword: LDS address = M0 + 4 * lane): no register is held while it
    // is in flight, and the compiler does not wait for it -- issue() and the explicit vmcnt wait in front of a slot's
    // first read keep the count: a slot's load is complete once at most TOK_RING - 1 younger loads are outstanding.
    auto fetch = [&]() {
        const uint32_t g = gnext;
        const uint32_t rel = pfirst - g;
        uint64_t bm = __ballot(rel < 64u), m = 0;
        const uint32_t inside = (uint32_t)__popcll(bm);
        while (bm) {
            const uint32_t k = (uint32_t)__ffsll((long long)bm) - 1u;
            m |= 1ull << rdlane(rel, k);
            bm &= bm - 1ull;
        }
        // pieces starting at or before lane t of the group: before + bits 0..t of m = before + (m & 1) + bits below t of m >> 1
        const uint32_t base = before + (uint32_t)(m & 1ull) - 1u;
        const uint64_t m1 = m >> 1;
        const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, base));
        const uint32_t d = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)pdelta);
        const uint32_t t = min(g + lane, ntok - 1u);  // (lanes behind the batch's end read its last token again)
        before += inside;
        gnext = g + 64u;
        const uint32_t krow = (t + d) & 0xffffu;  // the token's index in its lane's row
        const uint32_t *src = grow + ((d >> 16) + row_word(krow));
        const uint32_t dst = tok_lds + 256u * slot_w;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(dst)
                     : "memory");
        slot_w = (slot_w + 1u) & (TOK_RING - 1u);
    };
    WSYNC();  // the piece descriptions are in LDS; whatever used the chunk state's place before is done
    uint32_t c0 = 0;      // tokens executed so far
    uint32_t slot_r = 0;  // ring slot of the group at c0
#pragma unroll
    for (uint32_t k = 0; k < TOK_RING; k++) fetch();
    // ---- chunk state (every piece of code below exists once: no closures, everything stays in registers)
    bool fresh = true;  // a chunk starts with the next group
    uint32_t mis = 0, run = 0, nq = 0, qh = 0, prod0 = 0, cstart = 0;
    uint8_t *base = gout;
    bool glob_ok = false, inflight = false;
    Round R;
    for (;;) {
        if (fresh) {
            STAT_ADD(13, 1);
            mis = (uint32_t)((uintptr_t)(gout + opos) & 3u);
            base = gout + opos - mis;  // byte x of the chunk lives at base[x]; base is dword aligned
            glob_ok = cap - opos >= 16u;
            run = mis;
            prod0 = opos - mis;  // output bytes in front of offset 0
            nq = qh = 0;
            cstart = c0;
            fresh = false;
        }
        static_assert(TOK_RING == 4, "the wait below counts three younger ring loads");
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // the load into slot_r has landed
        const uint32_t t = C.tok[64u * slot_r + lane];
        // token: [8:0] literal byte or match length, [9] match, [25:10] distance - 1
        // (a literal's byte is the token's low byte as it stands; length and distance are only formed where a match is queued)
        const bool ismatch = (t & 512u) != 0;
        const uint32_t len = ismatch ? t & 0x1ffu : 0u;
        const uint32_t val = ismatch ? __builtin_amdgcn_ubfe(t, 10, 16) + 1u : t;  // distance, or the literal (low byte)
        const uint32_t left = ntok - c0;
        uint32_t olen = ismatch ? len : 1u;
        olen = lane < left ? olen : 0u;  // (only the last group of a batch has lanes behind the end)
        const uint32_t incl = wave_incl_scan(olen);
        const uint32_t start = run + incl - olen;
        const uint32_t total = rdlane(incl, 63u);
        const uint64_t lenm = __ballot(len != 0);
        uint32_t nacc;
        bool ending, too_far = false;
        // The common group: 64 tokens that all fit the chunk, no distance reaching in front of the output's first byte.
        // (a distance cannot reach in front of the output once 32 KiB of it exist)
        if (left >= 64u && run + total <= CHUNK_BYTES && (prod0 >= 32768u || !__any(len != 0 && val > prod0 + start))) {
            asm volatile("; the common group" ::: "memory");  // (keeps the compiler from folding this path into the general one below)
            if (len == 0) img[start] = (uint8_t)val;
            else {
                const uint32_t qi = __builtin_amdgcn_mbcnt_hi((uint32_t)(lenm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lenm, nq)) & (MQ_CAP - 1u);
                C.mq[qi] = make_uint2(start, len | (val << 16));
            }
            nq += (uint32_t)__popcll(lenm);
            c0 += 64u;
            run += total;
            nacc = 64u;
            ending = c0 >= ntok;
        } else {
            const uint32_t nvalid = left < 64u ? left : 64u;
            // stops: the first token that does not fit the chunk, or with an "invalid distance too far back" (the distance
            // reaches before the first output byte)
            const uint64_t validm = nvalid == 64u ? ~0ull : (1ull << nvalid) - 1ull;
            const uint64_t nofit = __ballot(incl > CHUNK_BYTES - run) & validm;
            const uint64_t badm = __ballot(val > prod0 + start) & lenm & validm;
            const uint64_t stopm = nofit | badm;
            nacc = stopm ? (uint32_t)__ffsll((long long)stopm) - 1u : nvalid;
            // a group that does not fit is left whole to the next chunk (the prefetched group stays the right one) unless the
            // chunk is empty (long matches: 64 tokens can be 16 KB) or the stop is an error
            too_far = stopm != 0 && ((badm & ~nofit) >> nacc) & 1ull;
            if (stopm && !too_far && run != mis) nacc = 0;
            if (lane < nacc && len == 0) img[start] = (uint8_t)val;
            const uint64_t mm = lenm & (nacc == 64u ? ~0ull : (1ull << nacc) - 1ull);
            if (lane < nacc && len != 0) {
                const uint32_t qi = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, nq)) & (MQ_CAP - 1u);
                C.mq[qi] = make_uint2(start, len | (val << 16));
            }
            nq += (uint32_t)__popcll(mm);
            c0 += nacc;
            if (stopm) {
                if (nacc) run = rdlane(start, nacc);
            } else {
                run += total;
            }
            ending = stopm != 0 || c0 >= ntok;  // the chunk ends with this group
        }
        LSYNC();  // literals and queue entries are in LDS; the slot's tokens are in registers
        STAT_ACC(16);
        // ---- match rounds.  A round's source loads are started as soon as 64 matches are queued and it is finished
        // when the next 64 are (about three token groups later: the loads have landed by then), or at the chunk's end.
        if (nq - qh >= 64u || ending) {
            do {
                if (inflight) {
                    round_finish(R, C, base, mis STAT_ARG);
                    inflight = false;
                }
                const uint32_t pq = nq - qh;
                if (pq >= 64u || (ending && pq != 0)) {
                    const uint32_t nr = pq < 64u ? pq : 64u;
                    round_issue(R, C, base, mis, qh, nr, glob_ok);
                    qh += nr;
                    inflight = true;
                }
            } while (ending && inflight);
        }
        STAT_ACC(17);
        // ---- the token ring (behind the rounds: their waits do not hold a ring load that was only just issued)
        if (nacc == 64u) {
            fetch();  // refills the slot just read: the group TOK_RING ahead
            slot_r = (slot_r + 1u) & (TOK_RING - 1u);
        } else if (nacc != 0 && c0 < ntok) {  // a partly taken group: the stream moves by less than a group, the ring starts over
            gnext = c0;
            before = (uint32_t)__popcll(__ballot(pfirst < c0));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no load of the old ring is left to land on the new one
            slot_r = slot_w = 0;
#pragma unroll
            for (uint32_t k = 0; k < TOK_RING; k++) fetch();
        }
        if (!ending) continue;
        // ---- the finished chunk: offsets [mis, min(run, capacity)) of the LDS image go to HBM
        const uint32_t xcap = cap - (opos - mis);
        {
            const uint32_t xe = run < xcap ? run : xcap;
            for (uint32_t xq = 4u * lane; xq < xe; xq += 256u) {
                const uint32_t wv = C.out[xq >> 2];
                if (xq >= mis && xq + 4u <= xe) {
                    *(GAS uint32_t *)(base + xq) = wv;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (xq + k >= mis && xq + k < xe) ((GAS uint8_t *)base)[xq + k] = (uint8_t)(wv >> (8 * k));
                }
            }
        }
        WSYNC();  // the next chunk overwrites the image and may load these bytes from memory
        STAT_ACC(18);
        opos += run - mis;
        opos_ = opos;
        if (opos > cap) {
            opos_ = cap;
            status = CHIP_NEED_OUTPUT;
            ovf[0] = cstart;
            ovf[1] = mis;
            ovf[2] = xcap;
            return false;
        }
        if (too_far) {
            status = Z_DATA_ERROR;
            return false;
        }
        if (c0 >= ntok) return true;
        fresh = true;
    }
}
// ---- the walk of a super-round -----------------------------------------------------------------------------
constexpr size_t SCRATCH_WORDS = ROWS_WORDS;


// why a lane's chain ended
enum : uint32_t { R_JOIN = 1, R_LIMIT = 2, R_EOB = 3, R_NEED_INPUT = 4, R_BAD = 5 };

// One super-round: stages the input from the true token boundary B on, walks 64 chains, finds the true stream among them and
// leaves its description in L.fl.pk (ntok tokens in npieces pieces).  term_why / term_pos say how and where the stream ends.
// (pl: lane n's view of the stream's n-th piece, for the tokens kernel -- the same numbers that go to L.fl.pk)
struct PieceLane {
    uint32_t node;   // lane whose row holds the piece
    uint32_t e_a0;   // the piece's first token in that row
    uint32_t cnt;    // its tokens (0: no n-th piece)
    uint32_t first;  // stream index of its first token
};
__device__ CHIP_PHASE_FN uint32_t walk_round(WaveLds &L, const InWin &w, const uint32_t B_, const uint32_t end_bit_, uint32_t *rows_, const uint32_t xt_bits,
                                             uint32_t &ntok_out, uint32_t &term_why, uint32_t &term_pos, PieceLane &pl STAT_PARAM)
{
    const uint32_t lane = lane_id();
    const uint32_t B = rdfirst(B_), end_bit = rdfirst(end_bit_);
    GAS const uint32_t *const g32 = rdfirst_gptr(w.g32);
    const uint32_t total_dw = rdfirst(w.total_dw);
    GAS uint32_t *const myrow = rdfirst_gptr(rows_) + row_base(lane);
    uint16_t *const pool16 = (uint16_t *)L.pool;
    const uint32_t D0 = B >> 5;
    WSYNC();  // the phase before (header parse, previous flush) is done with the window's place
    {
        // all loads of the window go out before the first LDS store waits for one
        constexpr uint32_t PER_LANE = (WIN_DW + 63) / 64;
        uint32_t v[PER_LANE];
#pragma unroll
        for (uint32_t j = 0; j < PER_LANE; j++) {
            const uint32_t i = D0 + 64u * j + lane;
            v[j] = i < total_dw ? g32[i] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < PER_LANE; j++) {
            const uint32_t k = 64u * j + lane;
            if (k < WIN_DW) {
                L.w.win[k] = v[j];
                L.w.bm[k] = 0;
            }
        }
    }
    LSYNC();
    STAT_ACC(1);
    STAT_ADD(8, 1);
    // ---- walk: every lane decodes from its guessed start until it joins another lane's chain ----
    const uint32_t s0 = B + lane * S_BITS;
    const uint32_t own_end = s0 + S_BITS;
    uint32_t hard = own_end + xt_bits;  // a token is taken if it ends in front of this: the chain's limit, the input's end
    {
        const uint32_t sr_end = B + 64u * S_BITS;  // nobody to join behind the last segment
        hard = hard < sr_end ? hard : sr_end;
        hard = hard < end_bit ? hard : end_bit;
    }
    // Positions inside the walk count from the window's first bit (32 * D0): a position's byte address in the window and in the mark
    // bits is then a shift and a mask.  q = position of the next token; a step adds the token's length to it before it knows whether
    // the token can be taken -- a lane that drops out keeps the length (tl), and the token's start is q - tl.
    const uint32_t wbase = 32u * D0;
    const uint32_t own_end_r = own_end - wbase, hard_r = hard - wbase;
    uint32_t q = s0 - wbase, tl = 0, nst = 0;
    uint32_t jb = 0, z = 0;  // of the lane's last token: join bit, halt flags
    bool run = s0 < end_bit;
    // A token's work is straight-line code; a lane whose token cannot be taken (it joined another chain, met an end-of-block or
    // invalid code, or the token ends behind `hard`) drops out with jb / z / pn as that step left them: the reason is read off them
    // after the loop.
    auto token = [&](uint32_t &tok) -> bool {
        const uint32_t p = q;
        const uint32_t a = (p >> 3) & ~3u;
        const uint32_t bit = 1u << (p & 31u);
        const uint32_t mine = p < own_end_r ? bit : 0u;
        const uint32_t old = atomicOr((uint32_t *)((uint8_t *)L.w.bm + a), mine);
        jb = old & (bit - mine);  // a boundary of the segment's owner: from here on the two chains are one
        const uint32_t *wp = (const uint32_t *)((const uint8_t *)L.w.win + a);
        const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, p), hi = __builtin_amdgcn_alignbit(d2, d1, p);
        const uint32_t r = L.lit_root[lo & ((1u << LIT_ROOT) - 1u)];
        const uint32_t e = L.pool[(r >> 5) + __builtin_amdgcn_ubfe(lo, LIT_ROOT, r)];
        const uint32_t n1 = __builtin_amdgcn_ubfe(e, 10, 5);
        const uint32_t msk = (uint32_t)((int32_t)e >> 31);  // all ones for a length code
        const uint32_t w2 = __builtin_amdgcn_alignbit(hi, lo, n1);
        uint32_t m = L.dist_root[w2 & ((1u << DIST_ROOT) - 1u)] & msk;
        if (m & D_LONG) {  // a distance code of more than 8 bits (1 % of the matches): through its sub-table
            const uint32_t b16 = ((m >> 4) & 127u) | ((m >> 12) << 7);
            m = pool16[b16 + __builtin_amdgcn_ubfe(w2, DIST_ROOT, m)];
        }
        const uint32_t cl2 = m & 15u, eb2 = __builtin_amdgcn_ubfe(m, 4, 4);
        const uint32_t dm1 = (__builtin_amdgcn_ubfe(m, 8, 2) << eb2) + __builtin_amdgcn_ubfe(w2, cl2, eb2);
        z = (e & (F_HALT | F_INV)) | (m & D_BAD);
        tl = n1 + cl2 + eb2;
        q = p + tl;
        if ((jb | z) != 0 || q > hard_r) return false;
        const uint32_t v = __builtin_amdgcn_ubfe(lo, e, e >> 5) + __builtin_amdgcn_ubfe(e, 16, 9);
        tok = (((dm1 << 10) | 512u) & msk) | v;
        nst++;
        return true;
    };
    bool full = false;  // the lane's row is full
    while (__any(run)) {
        STAT_ADD(11, 1);
        uint32_t t4[4];  // (a slot whose token is not taken keeps whatever its registers hold: nothing reads a row behind its lane's count)
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "=v"(t4[k]));
        const uint32_t ng = nst;
        if (run) {
            if (nst + 4u > ROW_TOKENS) {
                run = false;
                full = true;
            } else {
                bool ok = token(t4[0]);
                if (ok) {
                    ok = token(t4[1]);
                    if (ok) {
                        ok = token(t4[2]);
                        if (ok) ok = token(t4[3]);
                    }
                }
                run = ok;
            }
        }
        if (nst > ng) *(GAS u32x4 *)(myrow + 8u * ng) = u32x4{t4[0], t4[1], t4[2], t4[3]};  // ng is a multiple of 4: row_word(ng)
    }
    // back to stream positions: a lane that dropped out stands behind the token it did not take
    const bool dropped = s0 < end_bit && !full;
    const uint32_t pn = q + wbase;                     // end of the lane's last token (taken or not)
    const uint32_t p = pn - (dropped ? tl : 0u);       // start of the token that was not taken / of the next one
    // why the lane's last token was not taken, in zlib's order of verdicts
    uint32_t why = R_LIMIT;  // the chain simply ends (limit reached, row full, or it never ran)
    if (s0 < end_bit && !full) {
        if (jb) why = R_JOIN;
        else if (pn > end_bit) why = R_NEED_INPUT;  // the token does not end inside the input
        else if (z) why = z == F_HALT ? (uint32_t)R_EOB : (uint32_t)R_BAD;
    }
    STAT_ADD(21, __popcll(__ballot(nst != 0)));
    STAT_ACC(2);
    // ---- the true stream: lane 0's chain, then the chain it joined from the join on, and so on ----
    // the lane a chain joined, and the index, in that lane's row, of the token that starts at the join: the boundaries the
    // segment's owner marked in front of it
    const uint32_t rel = p - B;
    const uint32_t jl = why == R_JOIN ? (__umul24(rel, SEG_MAGIC) >> SEG_SHIFT) & 63u : lane;
    uint32_t a_join = 0;
    if (__any(why == R_JOIN)) {
        const uint32_t sj = B + jl * S_BITS;            // the owner's first bit
        const uint32_t w0 = (sj >> 5) - D0;             // ... and its dword in the window
        const uint32_t lo_cut = sj & 31u;               // bits below this in the first word belong to the segment before
        const uint32_t nb = p - (sj & ~31u);            // marks in front of bit nb (counted from the first word's bit 0) count
#pragma unroll
        for (uint32_t t = 0; t < SEG_WORDS; t++) {
            uint32_t mw = L.w.bm[w0 + t];
            if (t == 0) mw &= ~0u << lo_cut;
            const int32_t k = (int32_t)nb - 32 * (int32_t)t;
            const uint32_t keep = k <= 0 ? 0u : k >= 32 ? 0xffffffffu : (1u << k) - 1u;
            a_join += __popc(mw & keep);
        }
        if (why != R_JOIN) a_join = 0;
    }
    // the n-th piece of the stream = the lane reached from lane 0 by n joins: powers of the join map by doubling, composed along
    // the bits of n (lanes: ds_bpermute, no memory); the map's fixed points are the chains that end the stream
    uint32_t node = 0;  // lane n: the lane holding the stream's n-th piece
    {
        uint32_t pw = jl;  // the join map to the power 2^r
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const uint32_t nx = lane_gather(pw, node);
            if ((lane >> r) & 1u) node = nx;
            pw = lane_gather(pw, pw);
        }
    }
    const uint32_t prev = wave_shr1(node);                    // the piece before (lane 0: none)
    const bool fresh = lane == 0 || node != prev;               // lanes behind the stream's end repeat its last piece
    const uint32_t e_nst = lane_gather(nst, node);
    const uint32_t g_a0 = lane_gather(a_join, prev);  // (every lane takes part: a lane that is switched off cannot be read)
    const uint32_t e_a0 = lane == 0 ? 0u : g_a0;      // where the stream enters the piece
    const uint32_t cnt = (fresh && e_nst > e_a0) ? e_nst - e_a0 : 0u;
    const uint32_t incl = wave_incl_scan(cnt);
    const uint64_t nonempty = __ballot(cnt != 0);
    pl.node = node;
    pl.e_a0 = e_a0;
    pl.cnt = cnt;
    pl.first = incl - cnt;
    WSYNC();  // the walk's LDS reads are done (the flush's state takes the window's place)
    if (cnt) {
        const uint32_t k = (uint32_t)__popcll(nonempty & lanemask_lt()), first = incl - cnt;
        L.fl.pk[2u * k] = first;
        L.fl.pk[2u * k + 1u] = ((e_a0 - first) & 0xffffu) | (row_base(node) << 16);  // row token = stream index + (a0 - first)
    }
    const uint32_t kz = rdlane(node, 63);  // the chain the stream ends in
    term_why = rdlane(why, kz);
    term_pos = rdlane(p, kz);
    ntok_out = rdlane(incl, 63u);
    WSYNC();
    STAT_ACC(3);
    STAT_ADD(9, __popcll(nonempty));
    STAT_ADD(10, ntok_out);
    return (uint32_t)__popcll(nonempty);
}


// ---- tokens kernel only: a unit's token stream in the arena, its record -------------------------------------------------------
// (see chip_internal.h for the record's layout; every member is wave-uniform)
struct Emit {
    uint32_t *arena = nullptr;  // token arena, `arena_words` long, handed out in pieces of PIPE_ARENA_WORDS through counters[1]
    uint32_t arena_words = 0;
    uint32_t *counters = nullptr;
    uint32_t cur = 0, end = 0;  // the wave's piece of the arena: next free word, end (kept from unit to unit)
    bool exhausted = false;     // the arena has run out: every further unit of this wave takes the one-kernel path
    uint32_t *rec_base = nullptr;  // the records, PIPE_REC_WORDS per unit
    uint32_t *rec = nullptr;    // the unit's record
    uint32_t nseg = 0;
    uint32_t ext_off = 0, ext_n = 0;  // the open run of tokens (ext_n == 0: none)
    bool fail = false;          // the unit does not fit the record / the arena
};

// The tokens of a walk round's true stream, piece by piece (lane n copies the n-th piece from the row it lies in), to dst[0 .. ntok)
// in stream order.  A row holds four tokens per 16 bytes (row_word): a piece is up to three single tokens, whole groups of four, up
// to three single tokens.  The loads go out in batches (the six singles and four groups, then four groups at a time) before the
// first store waits for one: the copy costs a handful of memory round trips, not one per group.
__device__ __forceinline__ void copy_pieces(uint32_t *dst_, const uint32_t *rows_, const PieceLane &pl)
{
    GAS uint32_t *const dst = (GAS uint32_t *)dst_;
    GAS const uint32_t *const row = (GAS const uint32_t *)rdfirst_gptr(rows_) + row_base(pl.node);
    const uint32_t k0 = pl.e_a0, d0 = pl.first, cnt = pl.cnt;
    uint32_t hk = (4u - (k0 & 3u)) & 3u;
    hk = hk < cnt ? hk : cnt;                        // single tokens in front
    const uint32_t kb = k0 + hk, nb4 = (cnt - hk) >> 2, tk = (cnt - hk) & 3u;  // first whole group, whole groups, single tokens behind
    const uint32_t kt = kb + 4u * nb4, db = d0 + hk, dt = db + 4u * nb4;
    uint32_t h[3] = {0, 0, 0}, t[3] = {0, 0, 0};
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        if (j < hk) h[j] = row[row_word(k0 + j)];
        if (j < tk) t[j] = row[row_word(kt + j)];
    }
    for (uint32_t i = 0; __any(i < nb4); i += 4u) {
        u32x4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            v[j] = u32x4{0, 0, 0, 0};
            if (i + j < nb4) v[j] = *(GAS const u32x4 *)(row + 8u * (kb + 4u * (i + j)));  // row_word(k) = 8 k for k = 0 mod 4
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
            if (i + j < nb4) *(GAS u32x4_a4 *)(dst + db + 4u * (i + j)) = v[j];
    }
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        if (j < hk) dst[d0 + j] = h[j];
        if (j < tk) dst[dt + j] = t[j];
    }
}

__device__ __forceinline__ void emit_segment(Emit &E, uint32_t w0, uint32_t w1)
{
    if (E.nseg >= PIPE_MAXSEG) {
        E.fail = true;
        return;
    }
    if (lane_id() == 0) {
        E.rec[8u + 2u * E.nseg] = w0;
        E.rec[9u + 2u * E.nseg] = w1;
    }
    E.nseg++;
}

__device__ __forceinline__ void emit_close(Emit &E)
{
    if (E.ext_n) emit_segment(E, E.ext_off, E.ext_n);
    E.ext_n = 0;
}

// The ntok tokens of a walk round's true stream, piece by piece (lane n copies the n-th piece from the row it lies in), appended
// to the unit's token stream.  False: no room.
__device__ CHIP_PHASE_FN bool emit_round(Emit &E, const uint32_t *rows_, const PieceLane &pl, const uint32_t ntok_)
{
    const uint32_t ntok = rdfirst(ntok_);
    if (ntok == 0) return true;
    if (E.exhausted || E.fail) {
        E.fail = true;
        return false;
    }
    if (E.cur + ntok > E.end) {  // (ntok <= 64 rows x ROW_TOKENS = a whole piece)
        emit_close(E);
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(&E.counters[1], PIPE_ARENA_WORDS);
        base = rdfirst(base);
        if (base > E.arena_words || E.arena_words - base < PIPE_ARENA_WORDS) {
            E.exhausted = true;
            E.fail = true;
            return false;
        }
        E.cur = base;
        E.end = base + PIPE_ARENA_WORDS;
    }
    if (E.ext_n == 0) E.ext_off = E.cur;
    copy_pieces(rdfirst_ptr(E.arena) + E.cur, rows_, pl);
    E.cur += ntok;
    E.ext_n += ntok;
    return true;
}

// ---- a block's tokens: super-rounds of walk, path resolve, execution ------------------------------------
// Decode the tokens of one deflate block from bit `pos` on (tables are in LDS), executing them into gout.  On return `pos` is
// behind the end-of-block code (status stays ST_RUNNING) or status holds the reason decoding stopped.
// PIPE (tokens kernel): the true stream is not executed but appended to the unit's token stream (E); a unit that does not fit
// leaves with status CHIP_NEED_OUTPUT and E.fail set.
template <bool PIPE>
__device__ CHIP_PHASE_FN void decode_block(WaveLds &L, InWin &w, uint32_t &pos, const uint32_t end_bit, uint8_t *gout, uint32_t &opos,
                                           const uint32_t cap, int32_t &status, uint32_t *rows, const uint32_t xt_bits, const uint32_t eob_len,
                                           const uint32_t flags, Emit &E STAT_PARAM)
{
    pos = rdfirst(pos);
    opos = rdfirst(opos);
    for (;;) {
        if (pos >= end_bit) {
            status = CHIP_NEED_INPUT;
            return;
        }
        STAT_T0();
        uint32_t ntok = 0, why = 0, tpos = 0;
        PieceLane pl;
        const uint32_t npk = walk_round(L, w, pos, end_bit, rows, xt_bits, ntok, why, tpos, pl STAT_ARG);
        int32_t st2 = ST_RUNNING;
        bool flushed = true;
        uint32_t ovf[3] = {0, 0, 0};
        if constexpr (PIPE) {
            if (npk && !emit_round(E, rows, pl, ntok)) {
                flushed = false;
                st2 = CHIP_NEED_OUTPUT;
            }
        } else {
#ifdef CHIP_EXP_NO_FLUSH  // ablation (no output): the walk alone -- header, tables, walk, path resolve, token rows
            asm volatile("" ::"s"(npk), "s"(ntok));
#else
            if (npk) flushed = flush_tokens(L, rows, ntok, npk, gout, opos, cap, st2, ovf STAT_ARG);
#endif
        }
        w.win0 = 0xffffffffu;  // the phases used the header window's place
        STAT_ACC(20);
        if (!flushed) {
            status = st2;
            // CHIP_F_COMPU_STATUS: zlib's position when the output filled (else the position stays at the round's start)
            if constexpr (!PIPE)
                if ((flags & F_COMPU_STATUS) && st2 == CHIP_NEED_OUTPUT) pos = overflow_bit(L, w, rows, ntok, npk, ovf[0], ovf[1], ovf[2], pos);
            return;
        }
        if (why == R_NEED_INPUT) {
            status = CHIP_NEED_INPUT;
            return;
        }
        if (why == R_EOB) {
            pos = tpos + eob_len;
            return;
        }
        if (why != R_LIMIT || tpos <= pos) {  // invalid code (a super-round always gets past its first token otherwise)
            status = Z_DATA_ERROR;
            return;
        }
        pos = tpos;  // on from where the true stream stopped
    }
}

#ifndef CHIP_WAVES_PER_SIMD
#define CHIP_WAVES_PER_SIMD 5  // (18 waves per CU: two SIMDs hold five; 96 lane registers)
#endif
// one unit, start to finish, by the calling wave; scratch = the wave's token rows in HBM
// PIPE (tokens kernel): nothing is executed and no result is written -- the unit's tokens and stored runs go to its record (E); a
// unit that does not reach the end of its stream, or does not fit, is put on the fallback list for inflate_kernel.
template <bool PIPE>
__device__ __attribute__((always_inline)) void inflate_unit(const BatchArgs &a, const uint32_t u, WaveLds &L, uint32_t *scratch, Emit &E, uint32_t *fallback)
{
    const uint32_t lane = lane_id();

    const uint8_t *gin = a.in_base + a.in_off[u];
    const uint32_t in_len = a.in_len[u];
    uint8_t *gout = PIPE ? nullptr : a.out_base + a.out_off[u];
    const uint32_t cap = PIPE ? 0xffffffffu : a.out_cap[u];
    if constexpr (PIPE) {
        E.rec = rdfirst_ptr(E.rec_base + (size_t)u * PIPE_REC_WORDS);
        E.nseg = 0;
        E.ext_n = 0;
        E.fail = false;
    }

    InWin w;
    const uint32_t mis = (uint32_t)((uintptr_t)gin & 3u);
    w.g32 = (const uint32_t *)(gin - mis);
    w.total_dw = (mis + in_len + 3u) >> 2;
    w.win0 = 0xffffffffu;
    const uint32_t start_bit = mis * 8u;
    const uint32_t end_bit = start_bit + in_len * 8u;

    uint32_t pos = start_bit;
    uint32_t opos = 0;
    int32_t status = ST_RUNNING;
    bool last = false;
    int tables = 0;  // 0 none, 1 fixed, 2 dynamic
    uint32_t eob_len = 0;   // length of the block's end-of-block code

    STAT_DECL;
    STAT_T0();
    uint32_t wrap = 0;
    int32_t format = a.format;
    uint32_t ck_bit = 0, ck_opos = 0;  // last block boundary reached (streaming decoder: where the next call resumes)
    bool resumed = false;
    uint32_t run_check = 0, run_cov = 0, out_dropped = 0;  // running trailer check: value, stream bytes covered; bytes dropped in front
    if (!PIPE && a.resume) {
        const uint32_t *rs = a.resume + RESUME_WORDS * u;
        const uint32_t r0 = rs[0], r1 = rs[1], r2 = rs[2];
        if (r0 != 0 && r0 <= in_len * 8u && r1 <= cap) {
            resumed = true;
            pos = start_bit + r0;
            opos = r1;
            wrap = r2 & 3u;
            ck_bit = r0;
            ck_opos = r1;
            run_check = rs[3];
            run_cov = rs[4];
            out_dropped = rs[5];
        }
    }
    if (!resumed && format != CHIP_FMT_DEFLATE) {
        uint32_t hdr = 0;
        status = parse_wrapper((LDS_AS uint32_t *)&L, gin, in_len, format, wrap, hdr);
        status = (int32_t)rdfirst((uint32_t)status);
        wrap = rdfirst(wrap);
        pos += rdfirst(hdr) * 8u;
    }
    if (status == ST_RUNNING) win_load(L, w, pos >> 5);

    __builtin_amdgcn_s_setprio(2);
    while (status == ST_RUNNING) {
        if (last) {
            status = CHIP_FINISHED;
            break;
        }
        ck_bit = pos - start_bit;
        ck_opos = opos;
        win_ensure(L, w, pos);
        if (pos + 3 > end_bit) {
            status = CHIP_NEED_INPUT;
            break;
        }
        uint32_t lo, hi;
        win_bits_uniform(L, w, pos, lo, hi);
        last = lo & 1u;
        uint32_t type = (lo >> 1) & 3u;
        pos += 3;
        if (type == 0) {
            // stored block, RFC 1951 sec. 3.2.4
            pos = (pos + 7u) & ~7u;
            if (pos + 32 > end_bit) {
                status = CHIP_NEED_INPUT;
                break;
            }
            win_ensure(L, w, pos);
            win_bits_uniform(L, w, pos, lo, hi);
            uint32_t blen = lo & 0xffffu, nlen = lo >> 16;
            if (blen != (nlen ^ 0xffffu)) {
                status = Z_DATA_ERROR;
                break;
            }
            pos += 32;
            uint32_t avail = (end_bit - pos) >> 3;
            uint32_t room = cap - opos;
            uint32_t ncopy = blen < avail ? blen : avail;
            if (ncopy > room) ncopy = room;
            if constexpr (PIPE) {
                if (ncopy == blen && blen) {  // a stored run of the record (a short one goes the one-kernel way: status below)
                    emit_close(E);
                    emit_segment(E, (pos >> 3) - mis, blen | PIPE_SEG_STORED);
                }
            } else {
                wave_copy_stored(gout + opos, w.g32, w.total_dw, pos >> 3, ncopy);
                opos += ncopy;
            }
            pos += ncopy * 8u;
            if (ncopy < blen) {
                // zlib reports Z_OK here; compu calls it NeedInput when no input is left (mod.rs:476-479)
                status = ncopy == avail ? CHIP_NEED_INPUT : CHIP_NEED_OUTPUT;
                break;
            }
            continue;
        }
        if (type == 3) {
            status = Z_DATA_ERROR;
            break;
        }
        uint32_t nlen = 288, ndist = 32;
        bool build = true;
        if (type == 1) {
            if (tables != 1) {  // RFC 1951 sec. 3.2.6 (a run of fixed blocks keeps its tables)
                for (uint32_t s = lane; s < 320; s += 64) L.hdr.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
                WSYNC();
                tables = 1;
            } else build = false;
        } else {
            // dynamic block header, RFC 1951 sec. 3.2.7
            tables = 2;
            if (pos + 14 > end_bit) {
                status = CHIP_NEED_INPUT;
                break;
            }
            win_ensure(L, w, pos);
            win_bits_uniform(L, w, pos, lo, hi);
            nlen = (lo & 31u) + 257;
            ndist = ((lo >> 5) & 31u) + 1;
            const uint32_t ncode = ((lo >> 10) & 15u) + 4;
            pos += 14;
            if (nlen > 286 || ndist > 30) {
                status = Z_DATA_ERROR;
                break;
            }
            if (pos + 3 * ncode > end_bit) {
                status = CHIP_NEED_INPUT;
                break;
            }
            win_ensure(L, w, pos, 128);
            if (lane < 19) L.hdr.lens[lane] = 0;
            WSYNC();
            if (lane < ncode) {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint32_t l2, h2;
                win_bits(L, w, pos + 3 * lane, l2, h2);
                L.hdr.lens[order[lane]] = (uint8_t)(l2 & 7u);
            }
            WSYNC();
            pos += 3 * ncode;
            if (build_cl_table(L)) {
                status = Z_DATA_ERROR;
                break;
            }
            // Code lengths of the two alphabets, 64 bit positions per round: lane i decodes the code-length symbol
            // that would start at bit pos + i; the symbols really present are the chain 0 -> 0 + bits(0) -> ...,
            // followed with scalar readlanes; a prefix sum of the run lengths along the chain gives every symbol
            // its place in lens[], and a running maximum hands each "repeat previous" symbol the last explicit
            // length in front of it.  (zlib reads these one by one; errors keep its order: the first bad symbol.)
            uint32_t have = 0, total = nlen + ndist, prevlen = 0;
            win_ensure(L, w, pos, 316 * 14 + 160);
            for (uint32_t k = lane; k < 320; k += 64) L.hdr.lens[k] = 0;  // zero runs (17/18) then need no stores
            WSYNC();
            while (have < total) {
                const uint32_t bp = pos + lane;
                uint32_t lo, hi;
                win_bits(L, w, bp, lo, hi);
                const uint32_t e = L.hdr.cl_lut[lo & 127u];
                const uint32_t cl = e & 15u, sym = e >> 16;
                const uint32_t eb = sym < 16 ? 0u : sym == 16 ? 2u : sym == 17 ? 3u : 7u;
                const uint32_t tl = cl + eb;
                const uint32_t run = sym < 16 ? 1u : (sym == 18 ? 11u : 3u) + bfe(lo, cl, eb);
                const uint32_t nx = lane + tl;
                // The symbols really present start at 0, nx(0), nx(nx(0)), ...: lane n finds the n-th of these positions (powers of nx
                // by doubling, composed along the bits of n; 64 = the chain has left the 64 positions) and flags it.
                uint32_t nth = 0;
                {
                    uint32_t pw = nx < 64u ? nx : 64u;
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const uint32_t g1 = lane_gather(pw, nth & 63u), g2 = lane_gather(pw, pw & 63u);
                        if ((lane >> r) & 1u) nth = nth < 64u ? g1 : 64u;
                        pw = pw < 64u ? g2 : 64u;
                    }
                }
                uint8_t *const onflag = (uint8_t *)L.hdr.count;  // 64 bytes (the counts are not in use here)
                if (lane < 16) L.hdr.count[lane] = 0;
                LSYNC();
                if (nth < 64u) onflag[nth] = 1;
                LSYNC();
                const bool on = onflag[lane] != 0;
                const uint32_t c = on ? run : 0u;
                const uint32_t incl = wave_incl_scan(c);
                const uint32_t start = have + incl - c;
                const bool need = on && start < total;
                uint32_t bad = 0;  // zlib's checks, in its order, for the symbol this lane holds
                if (need && bp + tl > end_bit) bad = 1;
                else if (need && sym == 16 && start == 0) bad = 2;
                else if (need && sym >= 16 && start + run > total) bad = 2;
                const uint64_t badm = __ballot(bad != 0);
                const uint32_t fb = badm ? (uint32_t)__ffsll((long long)badm) - 1u : 64u;
                const bool ok = need && lane < fb;
                const uint32_t expl = sym < 16 ? sym : 0u;  // the length this symbol leaves as "previous"
                const uint32_t m = wave_incl_max_scan((ok && sym != 16) ? (((lane + 1u) << 8) | expl) : 0u);
                const uint32_t v = sym == 16 ? (m ? m & 0xffu : prevlen) : expl;
                if (ok && sym < 16) L.hdr.lens[start] = (uint8_t)sym;
                if (ok && sym == 16)
                    for (uint32_t j = 0; j < run; j++) L.hdr.lens[start + j] = (uint8_t)v;
                if (badm) {
                    status = rdlane(bad, fb) == 1 ? (int32_t)CHIP_NEED_INPUT : Z_DATA_ERROR;
                    break;
                }
                const uint64_t okm = __ballot(ok);
                const uint32_t lv = 63u - (uint32_t)__clzll((long long)okm);  // lane 0 is always ok here
                have = rdlane(start + run, lv);
                pos += lv + rdlane(tl, lv);
                prevlen = rdlane(v, lv);
            }
            if (status != ST_RUNNING) break;
            WSYNC();
            STAT_ACC(4);
            if (rdfirst(L.hdr.lens[256]) == 0) {
                status = Z_DATA_ERROR;
                break;
            }
        }
        if (build) {
            eob_len = rdfirst(L.hdr.lens[256]);
            uint32_t used = 0;
            const int r1 = build_litlen(L, L.hdr.lens, (int)nlen, used);
            const int r2 = r1 ? r1 : build_dist(L, L.hdr.lens + nlen, (int)ndist, used);
            if (r1 || r2) {  // an invalid set of code lengths (the pool cannot overflow: see POOL_WORDS)
                status = Z_DATA_ERROR;
                break;
            }
            STAT_ACC(5);
        }
        STAT_ACC(0);
        __builtin_amdgcn_s_setprio(0);
        decode_block<PIPE>(L, w, pos, end_bit, gout, opos, cap, status, scratch, tables == 1 ? XT_BITS_FIXED : XT_BITS, eob_len, a.flags, E STAT_ARG);
        __builtin_amdgcn_s_setprio(2);  // block headers and table builds are short dependent chains: ahead of the other waves' bulk work
        STAT_T0();
    }
    __builtin_amdgcn_s_setprio(0);
    STAT_ACC(0);
    if constexpr (PIPE) {
        // The record is complete when the stream ended at its last block and the trailer is there; the trailer's values are
        // checked by lz77_kernel, which has the bytes.  Everything else is inflate_kernel's: it reports what zlib would.
        uint32_t want = 0, isize = 0;
        uint32_t k = (pos - start_bit + 7u) >> 3;
        bool done = status == CHIP_FINISHED && !E.fail;
        if (done && wrap == 1) {
            if (in_len - k < 4) done = false;
            else want = ((uint32_t)gin[k] << 24) | ((uint32_t)gin[k + 1] << 16) | ((uint32_t)gin[k + 2] << 8) | gin[k + 3];
            k += 4;
        } else if (done && wrap == 2) {
            if (in_len - k < 8) done = false;
            else {
                want = gin[k] | ((uint32_t)gin[k + 1] << 8) | ((uint32_t)gin[k + 2] << 16) | ((uint32_t)gin[k + 3] << 24);
                isize = gin[k + 4] | ((uint32_t)gin[k + 5] << 8) | ((uint32_t)gin[k + 6] << 16) | ((uint32_t)gin[k + 7] << 24);
            }
            k += 8;
        }
        if (done) emit_close(E);
        done = done && !E.fail;
        if (lane == 0) {
            E.rec[0] = done ? PIPE_ST_TOKENS : PIPE_ST_FALLBACK;
            E.rec[1] = E.nseg;
            E.rec[2] = wrap;
            E.rec[3] = rdfirst(want);
            E.rec[4] = rdfirst(isize);
            E.rec[5] = k < in_len ? k : in_len;
            if (!done) fallback[atomicAdd(&E.counters[2], 1u)] = u;
        }
        return;
    }
    if (status == CHIP_FINISHED && wrap) {
        // trailer: gzip CRC-32 + ISIZE (little endian), zlib Adler-32 (big endian)
        uint32_t k = (pos - start_bit + 7u) >> 3;
        uint32_t need = wrap == 2 ? 8u : 4u;
        if (in_len - k < need && wrap == 1) {
            status = CHIP_NEED_INPUT;
        } else if (wrap == 1) {
            uint32_t want = ((uint32_t)gin[k] << 24) | ((uint32_t)gin[k + 1] << 16) | ((uint32_t)gin[k + 2] << 8) | gin[k + 3];
            const uint32_t c0 = run_cov - out_dropped;  // offset the running value (seed) covers up to; 0 and seed 1 for a whole stream
            if (rdfirst(wave_adler32(gout + c0, opos - c0, resumed ? run_check : 1u)) != rdfirst(want)) status = Z_DATA_ERROR;  // incorrect data check
            k += 4;
        } else if (in_len - k < 4) {
            status = CHIP_NEED_INPUT;
        } else {
            uint32_t want = gin[k] | ((uint32_t)gin[k + 1] << 8) | ((uint32_t)gin[k + 2] << 16) | ((uint32_t)gin[k + 3] << 24);
            const uint32_t c0 = run_cov - out_dropped;
            if (rdfirst(wave_crc32((LDS_AS uint32_t *)&L, gout + c0, opos - c0, resumed ? run_check : 0u)) != rdfirst(want)) status = Z_DATA_ERROR;  // incorrect data check
            else if (in_len - k < 8) status = CHIP_NEED_INPUT;
            else {
                uint32_t isize = gin[k + 4] | ((uint32_t)gin[k + 5] << 8) | ((uint32_t)gin[k + 6] << 16) | ((uint32_t)gin[k + 7] << 24);
                if (rdfirst(isize) != out_dropped + opos) status = Z_DATA_ERROR;  // incorrect length check
            }
            k += 8;
        }
        pos = start_bit + k * 8u;
    }
    STAT_ACC(6);
#ifdef CHIP_STATS
    if (a.stats && lane == 0)
        for (int k = 0; k < 24; k++) a.stats[(size_t)u * 24 + k] = st_[k];
#endif
    if (a.resume) {
        // the stream goes on in a later call: bring the running check up to the boundary it will resume from (the bytes in
        // front of it are final; the caller may drop them once it has handed them on)
        const bool cont = status == CHIP_NEED_INPUT || status == CHIP_NEED_OUTPUT;
        uint32_t *rs = a.resume + RESUME_WORDS * u;
        if (!resumed) run_check = wrap == 1 ? 1u : 0u;
        const uint32_t c0 = run_cov - out_dropped;
        if (cont && wrap && ck_bit != 0 && ck_opos > c0) {
            run_check = wrap == 1 ? wave_adler32(gout + c0, ck_opos - c0, run_check) : wave_crc32((LDS_AS uint32_t *)&L, gout + c0, ck_opos - c0, run_check);
            run_cov = out_dropped + ck_opos;
        }
        if (lane == 0) {
            rs[0] = cont ? ck_bit : 0u;
            rs[1] = cont ? ck_opos : 0u;
            rs[2] = wrap;
            rs[3] = run_check;
            rs[4] = run_cov;
            rs[5] = out_dropped;
        }
    }
    uint32_t used = (pos - start_bit + 7u) >> 3;
    if (used > in_len) used = in_len;
    if (a.flags & F_COMPU_STATUS) {
        // compu's reading of zlib's return code, src/decoder/mod.rs:475-483: Z_OK with avail_in == 0 is NeedInput whatever
        // else ran out, and a call that could not move (no input at all: Z_BUF_ERROR) is NeedOutput
        if (status == CHIP_NEED_OUTPUT && used == in_len) status = CHIP_NEED_INPUT;
        else if (status == CHIP_NEED_INPUT && in_len == 0) status = CHIP_NEED_OUTPUT;
    }
    if (lane == 0) {
        a.out_len[u] = opos;
        a.in_used[u] = status == CHIP_NEED_INPUT ? in_len : used;
        a.status[u] = status;
    }
}

// Persistent grid: each wave takes the next unit from *next_unit until the batch is exhausted.
__global__ __launch_bounds__(64, CHIP_WAVES_PER_SIMD) void inflate_kernel(BatchArgs a, uint32_t *scratch, uint32_t *next_unit)
{
    __shared__ WaveLds L;
    uint32_t *grow = scratch + (size_t)blockIdx.x * SCRATCH_WORDS;
    const uint32_t limit = a.sel_n ? *a.sel_n : a.n;
    Emit none;
    for (;;) {
        uint32_t i = 0;
        if (lane_id() == 0) i = atomicAdd(next_unit, 1u);
        i = rdfirst(i);
        if (i >= limit) break;
        const uint32_t u = a.sel ? rdfirst(a.sel[i]) : i;
        inflate_unit<false>(a, u, L, grow, none, nullptr);
        WSYNC();  // the next unit reuses the LDS
    }
}

// First kernel of the pipeline (chip_internal.h): the same persistent grid, but a unit's tokens go to the arena instead of being
// executed; lz77_kernel follows on the stream, then inflate_kernel over the fallback list.
__global__ __launch_bounds__(64, CHIP_WAVES_PER_SIMD) void tokens_kernel(BatchArgs a, uint32_t *scratch, PipeScratch p)
{
    __shared__ WaveLds L;
    uint32_t *grow = scratch + (size_t)blockIdx.x * SCRATCH_WORDS;
    const uint32_t limit = a.sel_n ? *a.sel_n : a.n;
    Emit E;
    E.arena = p.arena;
    E.arena_words = p.arena_words;
    E.counters = p.counters;
    E.rec_base = p.rec;
    for (;;) {
        uint32_t i = 0;
        if (lane_id() == 0) i = atomicAdd(&p.counters[0], 1u);
        i = rdfirst(i);
        if (i >= limit) break;
        const uint32_t u = a.sel ? rdfirst(a.sel[i]) : i;
        inflate_unit<true>(a, u, L, grow, E, p.fallback);
        WSYNC();  // the next unit reuses the LDS
    }
}

namespace {
// Token scratch and the unit counter of a launch, cached per (device, stream): launches on one stream
// run in order, so they can share a slot; different streams get their own.
struct LaunchSlot {
    uint32_t *scratch = nullptr;
    uint32_t *counter = nullptr;
    int blocks = 0;
    uint32_t *route = nullptr;  // routed batches: [0,1] list lengths, then two index lists of route_cap entries
    size_t route_cap = 0;
    // the two-kernel pipeline's scratch, sized by the largest batch launched here: unit records, fallback list, counters, token arena
    uint32_t *pipe_rec = nullptr, *pipe_fb = nullptr, *pipe_arena = nullptr;
    size_t pipe_units = 0, pipe_arena_words = 0;
    bool pipe_off = false;  // the arena could not be allocated: this slot stays with the one-kernel path
};
std::mutex g_slot_mu;
std::map<std::pair<int, hipStream_t>, LaunchSlot> g_slots;

// The slot of (current device, stream), with token scratch for min(n, resident waves) waves: a streaming decoder
// (batches of one) holds one 64 KiB slot, not the 270 MB a full grid needs; the scratch grows when a larger batch arrives.
// (caller holds g_slot_mu)
hipError_t slot_for(hipStream_t stream, uint32_t n, LaunchSlot &out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    LaunchSlot &sl = g_slots[{dev, stream}];
    static int max_blocks[64] = {0};  // resident waves of a full grid, per device
    const int di = dev < 64 ? dev : 63;
    if (!max_blocks[di]) {
        int per_cu = 0, cus = 0;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, inflate_kernel, 64, 0)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
#ifdef CHIP_EXP_PER_CU  // occupancy probe: a smaller persistent grid (waves per CU)
        per_cu = CHIP_EXP_PER_CU;
#endif
        max_blocks[di] = per_cu * cus;
        if (getenv("CHIP_DEBUG_GRID")) fprintf(stderr, "[chip] inflate_kernel: %d waves per CU x %d CUs resident (LDS %zu B per wave)\n", per_cu, cus, sizeof(WaveLds));
    }
    const int want = n < (uint32_t)max_blocks[di] ? (int)n : max_blocks[di];
    if (sl.blocks < want) {
        if (sl.scratch && (e = hipStreamSynchronize(stream)) != hipSuccess) return e;  // launches on the stream still use it
        (void)hipFree(sl.scratch);
        sl.scratch = nullptr;
        sl.blocks = 0;
        // a little headroom for batches that grow slowly (streaming objects stay at one wave)
        const int blocks = want <= 1 ? 1 : (want + want / 4 < max_blocks[di] ? want + want / 4 : max_blocks[di]);
        uint32_t *p = nullptr;
        if ((e = hipMalloc((void **)&p, (size_t)blocks * SCRATCH_WORDS * 4 + 256)) != hipSuccess) return e;
        sl.scratch = p;
        sl.counter = p + (size_t)blocks * SCRATCH_WORDS;
        sl.blocks = blocks;
    }
    out = sl;
    return hipSuccess;
}
}  // namespace

namespace {
// Pipeline scratch of the slot for a batch of n units (caller holds g_slot_mu).  The arena gets 32 Ki tokens (128 KB) per unit of the
// largest batch -- twice a 64 KiB unit's output, a 64 KiB unit of this class has 20-30 Ki tokens -- within a sixth of the device's
// memory; tokens_kernel sends the units that find no room to the one-kernel path.  False: not available, use the one-kernel path.
bool pipe_scratch(LaunchSlot &sl, hipStream_t stream, uint32_t n)
{
    if (sl.pipe_off) return false;
    if (n <= sl.pipe_units) return true;
    if (sl.pipe_rec && hipStreamSynchronize(stream) != hipSuccess) return false;  // launches on the stream still use it
    (void)hipFree(sl.pipe_rec);
    (void)hipFree(sl.pipe_arena);
    sl.pipe_rec = sl.pipe_fb = sl.pipe_arena = nullptr;
    sl.pipe_units = sl.pipe_arena_words = 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    const size_t units = (size_t)n + n / 4 + 16;
    size_t words = units * 32768;
    const size_t cap_words = (total_b / 6 < free_b / 2 ? total_b / 6 : free_b / 2) / 4;
    if (words > cap_words) words = cap_words;
    if (words > 0xffff0000ull) words = 0xffff0000ull;
    if (words < 4 * (size_t)PIPE_ARENA_WORDS) return false;
    uint32_t *rec = nullptr, *arena = nullptr;
    if (hipMalloc((void **)&rec, (units * (PIPE_REC_WORDS + 1) + 16) * 4) != hipSuccess) return false;
    if (hipMalloc((void **)&arena, words * 4) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(rec);
        sl.pipe_off = true;
        return false;
    }
    sl.pipe_rec = rec + 16;  // counters first
    sl.pipe_fb = rec + 16 + units * PIPE_REC_WORDS;
    sl.pipe_arena = arena;
    sl.pipe_units = units;
    sl.pipe_arena_words = words;
    return true;
}
bool pipe_enabled()
{
    static const bool on = [] {
        const char *e = getenv("CHIP_INFLATE_PIPE");  // 1: batches take the two-kernel pipeline (work in progress: not yet the faster path)
        return e && e[0] == '1';
    }();
    return on;
}
}  // namespace

namespace {
// (caller holds g_slot_mu)
hipError_t route_scratch_locked(hipStream_t stream, size_t n, uint32_t **sel_inflate, uint32_t **sel_zstd, uint32_t **counts)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    LaunchSlot &sl = g_slots[{dev, stream}];
    if (n > sl.route_cap) {
        // work queued on the stream may still read the old lists: let it finish before they go
        if (sl.route && (e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        (void)hipFree(sl.route);
        sl.route = nullptr;
        sl.route_cap = 0;
        const size_t cap = n + (n >> 2) + 1024;
        if ((e = hipMalloc((void **)&sl.route, (2 * cap + 4) * 4)) != hipSuccess) return e;
        sl.route_cap = cap;
    }
    *counts = sl.route;
    *sel_inflate = sl.route + 4;
    *sel_zstd = sl.route + 4 + sl.route_cap;
    return hipSuccess;
}
hipError_t launch_inflate_locked(const BatchArgs &a, hipStream_t stream);
}  // namespace

hipError_t route_scratch(hipStream_t stream, size_t n, uint32_t **sel_inflate, uint32_t **sel_zstd, uint32_t **counts)
{
    std::lock_guard<std::mutex> lk(g_slot_mu);
    return route_scratch_locked(stream, n, sel_inflate, sel_zstd, counts);
}

// A CHIP_FMT_DETECT batch: the router's lists and counters belong to (device, stream), so taking them, the counters' reset, the
// router and both decoders' launches are ONE critical section -- another host thread's routed batch on the same stream cannot put
// its reset or its router between this batch's router and this batch's decoders (which would then read the other batch's lists),
// nor free lists that these launches are about to read.  Replaces the routing of src/decoder/mod.rs:28-114 for a whole batch.
hipError_t launch_routed(const BatchArgs &a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_slot_mu);
    uint32_t *sel_i = nullptr, *sel_z = nullptr, *counts = nullptr;
    hipError_t e = route_scratch_locked(stream, a.n, &sel_i, &sel_z, &counts);
    if (e == hipSuccess) e = launch_route(a, sel_i, sel_z, counts, stream);
    BatchArgs ai = a, az = a;
    ai.format = CHIP_FMT_AUTO;
    ai.sel = sel_i;
    ai.sel_n = counts;
    az.format = CHIP_FMT_ZSTD;
    az.sel = sel_z;
    az.sel_n = counts + 1;
    if (e == hipSuccess) e = launch_inflate_locked(ai, stream);
    if (e == hipSuccess) e = launch_zstd_decode(az, 0, stream);
    return e;
}

hipError_t release_inflate_scratch()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_slot_mu);
    for (auto it = g_slots.begin(); it != g_slots.end();) {
        if (it->first.first == dev) {
            (void)hipFree(it->second.scratch);
            (void)hipFree(it->second.route);
            if (it->second.pipe_rec) (void)hipFree(it->second.pipe_rec - 16);
            (void)hipFree(it->second.pipe_arena);
            it = g_slots.erase(it);
        } else {
            ++it;
        }
    }
    return hipSuccess;
}

void release_inflate_scratch_of(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_slot_mu);
    auto it = g_slots.find({dev, stream});
    if (it != g_slots.end()) {
        (void)hipFree(it->second.scratch);
        (void)hipFree(it->second.route);
        if (it->second.pipe_rec) (void)hipFree(it->second.pipe_rec - 16);
        (void)hipFree(it->second.pipe_arena);
        g_slots.erase(it);
    }
}

hipError_t launch_inflate(const BatchArgs &a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    // One lock from the slot's lookup to the launch: a second host thread that launches a larger batch on the same stream may
    // free and reallocate the scratch in slot_for(); it must not do so between this thread's lookup and its launch (the
    // stream synchronisation in slot_for() only covers work that is already queued).  The counter reset and the kernel also
    // have to reach the stream back to back.
    std::lock_guard<std::mutex> lk(g_slot_mu);
    return launch_inflate_locked(a, stream);
}

namespace {
hipError_t launch_inflate_locked(const BatchArgs &a, hipStream_t stream)
{
    LaunchSlot sl;
    hipError_t e = slot_for(stream, a.n, sl);
    if (e != hipSuccess) return e;
    uint32_t blocks = a.n < (uint32_t)sl.blocks ? a.n : (uint32_t)sl.blocks;
#ifdef CHIP_EXP_EVEN_GRID  // probe: as many waves as give every wave the same number of units (no ragged last round)
    if (blocks) {
        const uint32_t per_wave = (a.n + blocks - 1) / blocks;
        blocks = (a.n + per_wave - 1) / per_wave;
    }
#endif
    // A batch (not a streaming decoder's call, which carries its state in a.resume) takes the two-kernel pipeline: tokens_kernel,
    // lz77_kernel, then inflate_kernel over the units those two have put on the fallback list (usually none).
    if (!a.resume && pipe_enabled()) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        LaunchSlot &ref = g_slots[{dev, stream}];
        if (pipe_scratch(ref, stream, a.n)) {
            PipeScratch p;
            p.rec = ref.pipe_rec;
            p.arena = ref.pipe_arena;
            p.arena_words = (uint32_t)ref.pipe_arena_words;
            p.counters = ref.pipe_rec - 16;
            p.fallback = ref.pipe_fb;
            if ((e = hipMemsetAsync(p.counters, 0, 16, stream)) != hipSuccess) return e;
            hipLaunchKernelGGL(tokens_kernel, dim3(blocks), dim3(64), 0, stream, a, sl.scratch, p);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if ((e = launch_lz77(a, p, stream)) != hipSuccess) return e;
            BatchArgs fb = a;
            fb.sel = p.fallback;
            fb.sel_n = p.counters + 2;
            hipLaunchKernelGGL(inflate_kernel, dim3(blocks), dim3(64), 0, stream, fb, sl.scratch, p.counters + 3);
            e = hipGetLastError();
            static const bool dbg = getenv("CHIP_PIPE_DEBUG") != nullptr;  // diagnostic: waits for the launches and prints the counters
            if (dbg && e == hipSuccess && hipStreamSynchronize(stream) == hipSuccess) {
                uint32_t c[4] = {0, 0, 0, 0};
                (void)hipMemcpy(c, p.counters, 16, hipMemcpyDeviceToHost);
                fprintf(stderr, "[chip pipe] units %u: taken %u, arena words %u of %u, fallback %u\n", a.n, c[0], c[1], p.arena_words, c[2]);
            }
            return e;
        }
    }
    if ((e = hipMemsetAsync(sl.counter, 0, 4, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(inflate_kernel, dim3(blocks), dim3(64), 0, stream, a, sl.scratch, sl.counter);
    return hipGetLastError();
}
}  // namespace

}  // namespace chip
