// Batched zstd frame decode for gfx950 (MI355X): one wavefront decodes one independent frame.
//
// Replaces, per frame, what compu reaches through ZSTD_decompressStream (src/decoder/zstd.rs:110) for
// a decoder built by Interface::zstd(opts) (src/decoder/zstd.rs:81-94): frame header, block loop,
// literals (raw / RLE / Huffman 1 or 4 streams / treeless), sequences (predefined / RLE / FSE /
// repeat tables, three interleaved FSE states on a backward bitstream), sequence execution with the
// repeat-offset history, and the XXH64 content checksum (RFC 8878).  No dictionaries (compu never
// loads one).
//
// Layout: the Huffman and FSE decode tables live in LDS; regenerated Huffman literals are parked at
// the END of the unit's output capacity (they are always consumed before the output cursor reaches
// them, as in libzstd's in-destination literal buffer); raw literals are read in place from the
// input; the match window is the output itself.
#include <cstddef>

#include "chip_internal.h"

namespace chip {

namespace {

constexpr uint32_t BLOCK_MAX = 128u * 1024u;

enum : int32_t { ZS_OK = 0 };

// sequence code tables, RFC 8878 sec. 3.1.1.3.2.1.1
__device__ const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const int8_t LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ const int8_t OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__device__ const int8_t ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};

// FSE decode entry: [5:0] symbol, [9:6] number of bits to read, [14:10] extra bits the symbol's code carries (LL / ML
// tables, filled in after the build), [31:16] baseline of the next state
template <int N>
struct FseTabN {
    uint32_t e[N];
    uint32_t al;
    uint32_t valid;
};
using FseTab = FseTabN<512>;   // literal-length / match-length tables (accuracy <= 9)
using FseTabOf = FseTabN<256>; // offset table (accuracy <= 8)
using FseTabWt = FseTabN<64>;  // Huffman-weight table (accuracy <= 6)
struct FseView {  // the builders work on any of the three sizes
    uint32_t *e, *al, *valid;
};
template <int N>
__device__ __forceinline__ FseView view(FseTabN<N> &t) { return FseView{t.e, &t.al, &t.valid}; }

struct alignas(16) ZLds {
    uint16_t huf[2048];  // [7:0] symbol, [11:8] code length
    FseTab ll, ml;
    FseTabOf of;
    uint32_t huf_bits, huf_valid;
    union {
        struct {  // table builds only (between them and the next build everything below is free for the phases)
            uint8_t weights[256];
            int16_t norm[64];
            uint16_t next[64];
            FseTabWt wt;  // FSE table of the Huffman weights
        };
        uint32_t seqwin[256];  // staged window of the sequence bitstream (read backward); literal decode: boundary rows (with xheads, xpar)
    };
    uint32_t xheads[64];   // copy phase: owner of every byte of a 256-byte step; table builds: running counts
    uint32_t xpar[192];    // copy phase: per-item parameters; state chain: the three states of every sequence of a chunk
    // sequence code -> baseline | extra bits << 24 for the codes that carry extra bits (LL 16.., ML 32..; below them the
    // baseline is the code itself, plus 3 for match lengths): read once per sequence
    uint32_t lltab[20], mltab[21];
};
static_assert(sizeof(ZLds) <= 11520, "fourteen waves per CU would fit the LDS (the registers allow twelve)");

// Streaming decoder only (BatchArgs::resume, one unit): checkpoint written after every completed block -- ZRES_HDR header words
// ([0] 1 + input bytes consumed (0 = none), [1] output bytes in the buffer, [2..4] repeat offsets, [5] flags: 1 checksum, 2 content
// size known, 4 all blocks done, [6,7] content size, [8,9] window, [10,11] output limit, [12,13] output bytes the HOST has dropped
// in front of the buffer (written by the host only), [14,15] output bytes the running XXH64 covers (a multiple of 32, counted from
// the frame's start), [16..23] its four accumulators) followed by the LDS image of the Huffman and FSE tables, which later blocks may
// reuse (treeless literals, repeat modes).  Between calls the host may drop input in front of [0] and output in front of
// min([1] - window, [14,15]) -- api.hip -- so a long frame is decoded in O(window) memory.
// (ZSAVE_WORDS, ZRES_HDR: chip_internal.h -- the host side sizes the checkpoint from them)
static_assert(offsetof(ZLds, weights) == ZSAVE_WORDS * 4, "checkpoint covers the decode tables");

// the unit's input seen as dwords (aligned down), addressed by absolute bit index
struct Bits {
    const uint32_t *g32;
    uint32_t total_dw;
};

__device__ __forceinline__ uint32_t rd32_at(const Bits &b, uint32_t bit)
{
    uint32_t i = bit >> 5;
    uint32_t d0 = i < b.total_dw ? b.g32[i] : 0u;
    uint32_t d1 = i + 1 < b.total_dw ? b.g32[i + 1] : 0u;
    return __builtin_amdgcn_alignbit(d1, d0, bit & 31u);
}

// backward bitstream over the absolute bit range [lo, lo + avail): the readers take bits from the top, MSB first
struct BackBits {
    uint32_t lo;
    int32_t avail;  // unread bits (negative once over-read)
};

// stream of `nbytes` bytes starting at absolute byte `byte0`; false if empty or its last byte is zero
__device__ __forceinline__ bool bb_init(const Bits &b, BackBits &s, uint32_t byte0, uint32_t nbytes)
{
    if (nbytes == 0) return false;
    uint32_t last = rd32_at(b, (byte0 + nbytes - 1) * 8u) & 0xffu;
    if (last == 0) return false;
    s.lo = byte0 * 8u;
    s.avail = (int32_t)((nbytes - 1) * 8u + (31u - (uint32_t)__clz((int)last)));
    return true;
}

// The sequence bitstream is read backward through a 1 KiB LDS window.  Once per sequence the 64 bits
// just below the read position are formed from three window dwords; the fields of the sequence are
// then cut from the top of that register pair, with one reload in the rare case a sequence needs more
// than 64 bits.  Bits below the stream start are whatever precedes it in the input: a read that
// reaches them drives `avail` negative, which the caller treats as corruption whatever the bits were.
struct SeqBits {
    uint32_t lo;    // absolute bit index of the stream start
    int32_t avail;  // unread bits of the stream (negative once over-read), not counting `used`
    int32_t win0;   // absolute dword index held in seqwin[0]
    uint64_t w;     // the 64 bits below lo + avail (bit 63 = the next bit)
    uint32_t used;  // bits already cut from w
};

__device__ __forceinline__ void sq_fill(ZLds &L, const Bits &b, SeqBits &s)
{
    WSYNC();
    s.win0 = ((int32_t)(s.lo + (uint32_t)s.avail) >> 5) + 2 - 256;
    uint32_t v[4];  // all four loads go out before the first LDS store waits for one
#pragma unroll
    for (int32_t r = 0; r < 4; r++) {
        const int32_t i = s.win0 + 64 * r + (int32_t)lane_id();
        v[r] = (i >= 0 && (uint32_t)i < b.total_dw) ? b.g32[i] : 0u;
    }
#pragma unroll
    for (int32_t r = 0; r < 4; r++) L.seqwin[64 * r + (int32_t)lane_id()] = v[r];
    WSYNC();
}

// commit the bits cut so far and form the next 64; restages the window when the read position (plus the
// <= 96 bits a sequence can take) comes near its lower end
__device__ __forceinline__ void sq_load(ZLds &L, const Bits &b, SeqBits &s)
{
    s.avail -= (int32_t)s.used;
    s.used = 0;
    const int32_t q = (int32_t)s.lo + s.avail - 64;  // lowest bit of the 64 (may lie below the stream: see above)
    if (((q - 96) >> 5) < s.win0) sq_fill(L, b, s);
    int32_t i = (q >> 5) - s.win0;
    i = i < 0 ? 0 : i;
    const uint32_t d0 = L.seqwin[i], d1 = L.seqwin[i + 1], d2 = L.seqwin[i + 2];
    const uint32_t sh = (uint32_t)q & 31u;
    s.w = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32) | __builtin_amdgcn_alignbit(d1, d0, sh);
}

__device__ __forceinline__ void sq_init(ZLds &L, const Bits &b, SeqBits &s, uint32_t lo, int32_t avail)
{
    s.lo = lo;
    s.avail = avail;
    s.used = 0;
    s.win0 = 0x7fffff00;  // forces the first staging
    sq_load(L, b, s);
}

__device__ __forceinline__ uint32_t sq_read(ZLds &L, const Bits &b, SeqBits &s, uint32_t n)  // n <= 32
{
    if (s.used + n > 64u) sq_load(L, b, s);
    const uint32_t v = n ? (uint32_t)((s.w << (s.used & 63u)) >> ((64u - n) & 63u)) : 0u;
    s.used += n;
    return v;
}

// (a << 2) + b in one instruction (the compiler emits a shift and an add)
__device__ __forceinline__ uint32_t lshl2_add(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t byte_at(const Bits &b, uint32_t byte) { return rd32_at(b, byte * 8u) & 0xffu; }

// ---- four-stream Huffman literals, 16 lanes per stream -------------------------------------------------
// Same idea as the inflate walk (inflate.hip): per super-round lane k of a stream starts HS_BITS * k bits below
// the stream's current read position (a guess, except for k = 0), decodes symbol boundaries from there, marks the
// boundaries inside its own segment in an LDS bitmap and, once below its segment, looks every new position up in
// the bitmap of the lane that owns the segment it is in; at the first hit the two chains are the same from there
// on.  The true chain is lane 0's up to its join, then the joined lane's from the join on, and so on; a popcount
// gives the index of the joined symbol and a prefix sum every piece's place in the literal buffer.  The walk keeps
// the symbols it decodes in scratch rows and the owned pieces are copied out of them (a second decode of exactly the
// owned symbols stands in when there is no room for the rows).  Huffman codes re-synchronise within a few symbols, so
// the 16 lanes cover 16 x 256 bits per round with ~70 serial steps instead of ~740.
#ifndef CHIP_HS_BITS
#define CHIP_HS_BITS 256
#endif
constexpr int HS_BITS = CHIP_HS_BITS;  // a lane's segment; a power of two makes segment and offset a shift and a mask
#ifndef CHIP_HXT_BITS
#define CHIP_HXT_BITS 512
#endif
constexpr int HXT_BITS = CHIP_HXT_BITS;
constexpr int HROW_WORDS = HS_BITS / 32;
static_assert(HROW_WORDS * 64 <= 256 + 64 + 192, "boundary rows live in seqwin + xheads + xpar");

struct HufStream {      // per lane: the stream its group of 16 lanes decodes
    uint32_t lo;        // absolute bit index of the stream's first bit
    uint32_t top;       // absolute bit index just above the next unread bit
    uint32_t want;      // symbols the stream must hold
    uint32_t done;      // symbols stored so far
    uint32_t out;       // index in gout of the stream's first literal
};

// Backward bit reader of the literal decoder: four symbols (at most 44 bits) are taken from a 64-bit view of the bits below the
// read position; the view comes out of a 16-byte window that was requested one trip earlier, when the position was known to within
// those 44 bits -- so a trip never waits for a load it has just issued, and there is no refill bookkeeping (round 4 found the old
// reader waiting for every refill: first behind a mask, then because a dword requested two symbols earlier had not arrived).
// Nothing is masked: bits outside the stream are whatever lies there -- a prefix code is decided by its own bits, and a symbol that
// needs more bits than the stream has left is refused by the caller's `r + nb > rend`, so they never decide anything.
typedef uint32_t zu32x4 __attribute__((ext_vector_type(4)));
typedef zu32x4 zu32x4_u __attribute__((aligned(1)));
struct HufWin {
    zu32x4 w;        // input bytes [base8 / 8, +16)
    uint32_t base8;  // absolute bit index of w's first bit
};
// The window for the trip after the one that starts at `pos` (absolute bit index just above the next unread bit): wherever that
// trip starts, within 44 bits below pos, its 64-bit view lies inside.  The address is clamped into the unit's dwords.
__device__ __forceinline__ HufWin huf_win_load(const Bits &b, uint32_t pos)
{
    int32_t a = ((int32_t)pos - 108) >> 3;
    int32_t last = (int32_t)(b.total_dw * 4u) - 16;
    last = last < 0 ? 0 : last;
    a = a < 0 ? 0 : (a > last ? last : a);
    HufWin h;
    h.w = *(const zu32x4_u *)((const uint8_t *)b.g32 + a);
    h.base8 = (uint32_t)a * 8u;
    return h;
}
// the 64 bits below pos, the next unread bit at bit 63
__device__ __forceinline__ uint64_t huf_win_view(const HufWin &h, uint32_t pos)
{
    const uint32_t o = (pos - 64u - h.base8) & 63u;
    const uint32_t t0 = __builtin_amdgcn_alignbit(h.w.y, h.w.x, o), t1 = __builtin_amdgcn_alignbit(h.w.z, h.w.y, o), t2 = __builtin_amdgcn_alignbit(h.w.w, h.w.z, o);
    const bool up = o >= 32u;
    return ((uint64_t)(up ? t2 : t1) << 32) | (up ? t1 : t0);
}

// Diagnostic build (-DCHIP_STATS, tools/stats_zstd.py): cycles per phase and trip counts of a frame, 24 words per unit at a.stats.
#ifdef CHIP_STATS
#define ZSTAT_PARAM , unsigned long long *zst
#define ZSTAT_ARG , zst
#define ZT_BEGIN(v) const unsigned long long v = __builtin_readcyclecounter()
#define ZT_END(i, v) (zst[i] += __builtin_readcyclecounter() - (v))
#define ZC(i, n) (zst[i] += (unsigned long long)(n))
#else
#define ZSTAT_PARAM
#define ZSTAT_ARG
#define ZT_BEGIN(v)
#define ZT_END(i, v)
#define ZC(i, n)
#endif

// The walk keeps what it decodes: every trip (four symbols) leaves one dword per lane in a scratch row -- [trip][lane], so a trip's
// 64 dwords are one 256-byte store -- and the owned pieces are then copied out of the rows instead of being decoded a second time.
// The scratch is the part of the frame's own output range that lies between what is written and the parked literals (free while
// a literal section decodes); without room for it (scr == nullptr), or in a round in which some lane walks more than HSCR_TRIPS trips
// (codes of a bit or two), the second decode below does the storing.
constexpr uint32_t HSCR_TRIPS = 32;                           // trips of a round the rows hold (128 symbols per lane)
constexpr uint32_t HSCR_BYTES = (HSCR_TRIPS + 2) * 256 + 4;   // (+ two rows that the copy's look-ahead may read, + alignment)

// Decodes the four streams described per lane by `hs` into gout.  Returns false on a corrupt stream.
__device__ bool huf_decode4(ZLds &L, const Bits &b, HufStream hs, uint8_t *gout, uint32_t *scr ZSTAT_PARAM)
{
    const uint32_t lane = lane_id(), k = lane & 15u, g0 = lane & ~15u;
    const uint32_t hbits = L.huf_bits;
    uint32_t *const rows = L.seqwin;                 // [word][lane]
    enum : uint32_t { H_IDLE = 0, H_JOIN = 1, H_LIMIT = 2, H_END = 3, H_BAD = 4 };
    bool bad = false;
    bool live = true;  // the stream still has symbols to find
    while (__any(live)) {
        ZC(11, 1);
        const uint32_t rend = hs.top - hs.lo;  // bits left in the stream
#pragma unroll
        for (int w = 0; w < HROW_WORDS; w++) rows[w * 64 + lane] = 0;
        WSYNC();
        // ---- walk
        const uint32_t r0 = k * HS_BITS;
        uint32_t rlim = r0 + HS_BITS + HXT_BITS;
        rlim = rlim < 16u * HS_BITS ? rlim : 16u * HS_BITS;
        uint32_t r = r0, nst = 0, reason = H_IDLE, jl = 64, rstop = r0;
        bool active = live && r0 < rend;
        if (live && k == 0 && rend == 0) reason = H_END;
        HufWin cur = huf_win_load(b, hs.top - (r0 < rend ? r0 : rend));
        uint32_t trip = 0;  // (uniform)
        uint32_t pend = 0;  // the trip before's symbols: stored at the top of the next trip, in front of that trip's window request, so
                            // that the wait for a window never has a younger store in front of it to wait for as well
        while (__any(active)) {
            ZC(12, 4);
            if (scr && trip && trip <= HSCR_TRIPS) scr[(trip - 1u) * 64u + lane] = pend;
            const uint32_t pos_it = hs.top - (r < rend ? r : rend);
            const HufWin nxt = huf_win_load(b, pos_it);  // (for the next trip: requested now, used then)
            uint64_t buf = huf_win_view(cur, pos_it);
            uint32_t word = 0;  // the trip's symbols (a lane that stops inside the trip leaves bytes nobody reads)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t rr = active ? r : r0;
                uint32_t seg, off;
                if constexpr ((HS_BITS & (HS_BITS - 1)) == 0) {
                    seg = rr / (uint32_t)HS_BITS;
                    off = rr % (uint32_t)HS_BITS;
                } else {
                    // rr / HS_BITS by a 24-bit multiply (full rate; exact while the error term stays under one: checked below)
                    constexpr uint32_t HS_MAGIC = ((1u << 20) + HS_BITS - 1) / HS_BITS;
                    static_assert(HS_MAGIC * HS_BITS - (1u << 20) < (1u << 20) / (16u * HS_BITS + HXT_BITS), "segment index by multiplication");
                    seg = __umul24(rr, HS_MAGIC) >> 20;
                    off = rr - __umul24(seg, (uint32_t)HS_BITS);
                }
                const uint32_t bit = 1u << (off & 31u);
                const uint32_t old = atomicOr(&rows[(off >> 5) * 64 + g0 + seg], (active && seg == k) ? bit : 0u);
                const bool joined = active && seg != k && (old & bit);
                const uint32_t e = L.huf[(uint32_t)(buf >> (64 - hbits))];
                const uint32_t nb = e >> 8;
                word |= (e & 0xffu) << (8 * u);
                uint32_t st = H_IDLE;
                st = r + nb > rend ? (uint32_t)H_BAD : st;
                st = joined ? (uint32_t)H_JOIN : st;
                const bool go = active && st == H_IDLE;
                const bool stop = active && !go;  // (selects, not branches: the step is straight-line code)
                reason = stop ? st : reason;
                jl = stop ? g0 + seg : jl;
                rstop = stop ? r : rstop;
                const uint32_t adv = go ? nb : 0u;
                nst += go ? 1u : 0u;
                r += adv;
                buf <<= adv;
                active = go && r < rlim && r < rend;
                const bool ran_out = go && !active;
                reason = ran_out ? (r >= rend ? (uint32_t)H_END : (uint32_t)H_LIMIT) : reason;
                rstop = ran_out ? r : rstop;
            }
            pend = word;
            trip++;
            cur = nxt;
        }
        if (scr && trip && trip <= HSCR_TRIPS) scr[(trip - 1u) * 64u + lane] = pend;
        const bool from_rows = scr && trip <= HSCR_TRIPS;
        ZC(18, from_rows ? 0 : 1);
        ZC(19, 1);
        WSYNC();
        // ---- the true chain of every stream: lane 0 of the group, then whatever it joined, ...
        const uint32_t nxt = reason == H_JOIN ? jl : 64u;
        uint32_t a_join = 0;
        if (reason == H_JOIN) {
            const uint32_t off = rstop - (jl & 15u) * HS_BITS;
#pragma unroll
            for (int w = 0; w < HROW_WORDS; w++) {
                int nbb = (int)off - 32 * w;
                nbb = nbb < 0 ? 0 : (nbb > 32 ? 32 : nbb);
                const uint32_t below = nbb >= 32 ? 0xffffffffu : ((1u << nbb) - 1u);
                a_join += __popc(rows[w * 64 + jl] & below);
            }
        }
        // The n-th piece of a stream = the lane reached from the group's first lane by n joins: powers of the join map by doubling,
        // composed along the bits of n (ds_bpermute: registers only, no barriers); the map's fixed points end the stream.  Lane k of
        // a group then works on the stream's k-th piece, whichever lane walked it: stream order, so one prefix sum places the pieces.
        auto gather = [](uint32_t x, uint32_t src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)x); };
        uint32_t node = g0;
        {
            uint32_t pw = nxt < 64u ? nxt : lane;
#pragma unroll
            for (int rd = 0; rd < 4; rd++) {
                const uint32_t nx = gather(pw, node);
                if ((k >> rd) & 1u) node = nx;
                pw = gather(pw, pw);
            }
        }
        const uint32_t prev = wave_shr1(node);  // the piece before (k = 0: none)
        const bool on = live && (k == 0 || node != prev);  // lanes behind the stream's end repeat its last piece
        const uint32_t e_nst = gather(nst, node);
        const uint32_t g_a0 = gather(a_join, prev), g_ps = gather(rstop, prev);  // (every lane takes part: a lane switched off cannot be read)
        const uint32_t a0 = k == 0 ? 0u : g_a0;
        const uint32_t pstart = k == 0 ? 0u : g_ps;  // offset below the stream top where this piece starts
        const uint32_t cnt = (on && e_nst > a0) ? e_nst - a0 : 0u;
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t below = (uint32_t)__shfl((int)incl, (int)((g0 - 1u) & 63u), 64);  // every lane takes part: the source lane must be active
        const uint32_t gbase = g0 ? below : 0u;
        const uint32_t first = incl - cnt - gbase;
        const uint32_t total = (uint32_t)__shfl((int)incl, (int)(g0 + 15u), 64) - gbase;
        // the stream's chain ends in the lane its last piece lies in
        const uint32_t lz = gather(node, g0 + 15u);
        const uint32_t rz = gather(reason, lz), sz = gather(rstop, lz);
        // ---- the owned symbols, stored: copied out of the walk's rows, or decoded once more
        {
            ZT_BEGIN(zt2);
            uint8_t *dst = gout + hs.out + hs.done + first;
            const uint32_t room = hs.want > hs.done + first ? hs.want - (hs.done + first) : 0u;  // never write past the stream's literals
            const uint32_t n = cnt < room ? cnt : room;
            struct __attribute__((packed, aligned(1))) HU32 {
                uint32_t v;
            };
#ifndef CHIP_EXP_NOHUF
            if (from_rows) {
                // piece = symbols [a0, a0 + n) of lane `node`'s chain: byte (j & 3) of row word j >> 2.  Eight output dwords per trip
                // from nine row words requested together; the funnel shift takes out the piece's misalignment in its row.
                const uint32_t w0 = on ? a0 >> 2 : 0u, sh = (a0 & 3u) * 8u;
                const uint32_t *const col = scr + (on ? node : lane);
                for (uint32_t i = 0; __any(i < n); i += 32) {
                    uint32_t rw[9];
#pragma unroll
                    for (uint32_t t = 0; t < 9; t++) {
                        uint32_t wi = w0 + (i >> 2) + t;
                        wi = wi < HSCR_TRIPS + 2u ? wi : HSCR_TRIPS + 1u;
                        rw[t] = col[wi * 64u];
                    }
#pragma unroll
                    for (uint32_t t = 0; t < 8; t++) {
                        const uint32_t word = __builtin_amdgcn_alignbit(rw[t + 1], rw[t], sh);
                        const uint32_t at = i + 4u * t;
                        if (at + 4 <= n) {
                            ((HU32 *)(dst + at))->v = word;
                        } else if (at < n) {
#pragma unroll
                            for (uint32_t q = 0; q < 3; q++)
                                if (at + q < n) dst[at + q] = (uint8_t)(word >> (8 * q));
                        }
                    }
                }
            } else {
            uint32_t pos2 = hs.top - (on ? pstart : 0u);
            HufWin cur2 = huf_win_load(b, pos2);
            // four symbols per trip, stored as one dword (at any alignment) instead of four scattered byte stores.  Every lane decodes in
            // every trip -- a lane past its count reads on into bits that are not its own (the window's address is clamped into the
            // buffer) and stores nothing -- so the trip is straight-line code.
            for (uint32_t i = 0; __any(i < n); i += 4) {
                const HufWin nxt2 = huf_win_load(b, pos2);
                uint64_t buf = huf_win_view(cur2, pos2);
                uint32_t word = 0;
#pragma unroll
                for (uint32_t t = 0; t < 4; t++) {
                    const uint32_t e = L.huf[(uint32_t)(buf >> (64 - hbits))];
                    word |= (e & 0xffu) << (8 * t);
                    buf <<= e >> 8;
                    pos2 -= e >> 8;
                }
                if (i + 4 <= n) {
                    ((HU32 *)(dst + i))->v = word;
                } else if (i < n) {
#pragma unroll
                    for (uint32_t t = 0; t < 3; t++)
                        if (i + t < n) dst[i + t] = (uint8_t)(word >> (8 * t));
                }
                cur2 = nxt2;
            }
            }
#endif
            ZT_END(2, zt2);
        }
#ifdef CHIP_DEBUG_HUF
        if ((lane & 15) == 0 || lane == 17) printf("l%u live%d rend%u r0%u reason%u rstop%u nst%u nxt%u a0%u on%d cnt%u first%u total%u lz%u rz%u sz%u want%u done%u\n", lane, (int)live, rend, r0, reason, rstop, nst, nxt, a0, (int)on, cnt, first, total, lz, rz, sz, hs.want, hs.done);
#endif
        if (live) {
            hs.done += total;
            hs.top -= sz;
            if (rz == H_BAD || hs.done > hs.want) bad = true;
            if (rz == H_END) {
                if (hs.done != hs.want) bad = true;
                live = false;
            } else if (rz != H_LIMIT) {
                live = false;  // H_BAD (or nothing found): the stream is corrupt
                bad = true;
            }
            if (bad) live = false;
        }
        WSYNC();
    }
#ifdef CHIP_DEBUG_HUF
    if ((lane & 15) == 0) printf("end l%u bad%d done%u want%u\n", lane, (int)bad, hs.done, hs.want);
#endif
    return !__any(bad);
}

// FSE table description (sec. 4.1.1) read forward from absolute byte `p0` (at most `n` bytes).
// Fills L.norm; returns bytes consumed or -1.  Uniform.
__device__ int fse_read_ncount(ZLds &L, const Bits &b, uint32_t p0, uint32_t n, int max_al, int max_sym, int &al_out, int &nsym_out)
{
    if (n < 1) return -1;
    uint32_t bit = rdfirst(p0 * 8u);
    const uint32_t endbit = rdfirst((p0 + n) * 8u);
    // The description is a few dozen bytes (at most 53 counts of at most ten bits, plus zero-run flags: under 80): lane i keeps dword i
    // of it and bits are read with readlane.  Everything here is the same in every lane and is kept on the scalar unit: a 64-bit
    // scalar shift cuts the bits (a vector funnel shift made the compiler hold the counts in vector registers, with an exec-mask region
    // per decision: 200 instructions per symbol), every lane stores the (same) count.
    const uint32_t d0 = bit >> 5;
    const uint32_t mydw = d0 + lane_id() < b.total_dw ? b.g32[d0 + lane_id()] : 0u;
    auto rdw = [&](uint32_t at) -> uint32_t {
        const uint32_t i = ((at >> 5) - d0) & 63u;
        const uint64_t ww = ((uint64_t)rdlane(mydw, (i + 1u) & 63u) << 32) | rdlane(mydw, i);
        return (uint32_t)(ww >> (at & 31u));
    };
    int al = (int)(rdw(bit) & 15u) + 5;
    bit += 4;
    if (al > max_al) return -1;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0, prev0 = 0;
    while (remaining > 1 && sym <= max_sym) {
        if (prev0) {
            int n0 = sym;
            while ((rdw(bit) & 3u) == 3u) {
                n0 += 3;
                bit += 2;
                if (bit > endbit + 7 || bit - 32u * d0 > 61u * 32u) return -1;
            }
            n0 += (int)(rdw(bit) & 3u);
            bit += 2;
            if (n0 > max_sym + 1) return -1;
            while (sym < n0) {
                L.norm[sym] = 0;
                sym++;
            }
            if (sym > max_sym) break;
        }
        int max = (2 * threshold - 1) - remaining, count;
        uint32_t bits = rdw(bit);
        if ((int)(bits & (uint32_t)(threshold - 1)) < max) {
            count = (int)(bits & (uint32_t)(threshold - 1));
            bit += (uint32_t)(nbits - 1);
        } else {
            count = (int)(bits & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            bit += (uint32_t)nbits;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        L.norm[sym] = (int16_t)count;
        sym++;
        prev0 = !count;
        while (remaining < threshold) {
            nbits--;
            threshold >>= 1;
        }
        if (bit > endbit + 7 || bit - 32u * d0 > 61u * 32u) return -1;  // (the second: past the staged dwords; no valid description gets there)
    }
    if (remaining != 1 || sym > max_sym + 1) return -1;
    al_out = al;
    nsym_out = sym;
    uint32_t used = ((bit + 7u) >> 3) - p0;
    if (used > n) return -1;
    return (int)used;
}

// FSE decoding table from normalized counts in L.norm (sec. 4.1.1), all lanes.  nsym <= 64: lane s keeps symbol s.
// The reference procedure walks the table with a fixed odd step and hands the visited cells (those above `high`, which
// hold the "less than one" symbols, are skipped) to the symbols in order.  Here every cell finds its symbol by itself:
// the step's inverse modulo the size tells when a position is visited, the skipped positions visited earlier are
// counted, and the symbol owning that cell number is found by bisection in the running sums of the counts.  The second
// half (bit count and baseline from the rank of a state among the states of its symbol) goes 64 states at a time.
__device__ int fse_build(ZLds &L, FseView t, int nsym, int al)
{
    WSYNC();
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t size = 1u << al, mask = size - 1u;
    const int c = (int)lane < nsym ? (int)L.norm[lane] : 0;
    const uint64_t negm = __ballot(c == -1);
    const uint32_t nneg = (uint32_t)__popcll(negm);
    const uint32_t cnt = c > 0 ? (uint32_t)c : 0u;
    const uint32_t incl = wave_incl_scan(cnt);
    if (rdlane(incl, 63) + nneg != size) return -1;  // the walk would not end on cell 0
    const uint32_t high = size - 1u - nneg;
    if (c == -1) t.e[size - 1u - (uint32_t)__popcll(negm & lt)] = lane;
    L.next[lane] = (uint16_t)(c == -1 ? 1u : cnt);
    L.xheads[lane] = incl;  // (copy-phase scratch, free here)
    const uint32_t step = (size >> 1) + (size >> 3) + 3u;
    uint32_t inv = step;  // Newton: correct bits double per round, 3 to begin with
#pragma unroll
    for (int i = 0; i < 3; i++) inv *= 2u - step * inv;
    const uint32_t kq = lane < nneg ? ((size - 1u - lane) * inv) & mask : 0xffffffffu;  // when the skipped positions are visited
    WSYNC();
    for (uint32_t p = lane; p <= high; p += 64) {
        const uint32_t k = (p * inv) & mask;
        uint32_t skipped = 0;
        for (uint32_t i = 0; i < nneg; i++) skipped += rdlane(kq, i) < k ? 1u : 0u;
        const uint32_t cell = k - skipped;
        uint32_t sym = 0;  // symbols whose running sum is <= cell
#pragma unroll
        for (uint32_t h = 32; h; h >>= 1)
            if (L.xheads[sym + h - 1u] <= cell) sym += h;
        t.e[p] = sym;
    }
    WSYNC();
    for (uint32_t u0 = 0; u0 < size; u0 += 64) {
        const uint32_t u = u0 + lane;
        const bool in = u < size;
        const uint32_t sy = in ? t.e[u] & 0xffu : 0u;
        uint64_t same = __ballot(in);  // lanes of this round with the same symbol
#pragma unroll
        for (uint32_t bit = 0; bit < 6; bit++) {
            const uint64_t bm = __ballot((sy >> bit) & 1u);
            same &= ((sy >> bit) & 1u) ? bm : ~bm;
        }
        if (in) {
            const uint32_t ns = (uint32_t)L.next[sy] + (uint32_t)__popcll(same & lt);
            const uint32_t nb = (uint32_t)al - (31u - (uint32_t)__clz((int)ns));
            t.e[u] = sy | (nb << 6) | ((((ns << nb) - size) & 0xffffu) << 16);
        }
        WSYNC();
        if (in && (same >> lane) == 1ull) L.next[sy] = (uint16_t)(L.next[sy] + (uint32_t)__popcll(same));  // the group's highest lane
        WSYNC();
    }
    if (lane == 0) {
        *t.al = (uint32_t)al;
        *t.valid = 1;
    }
    WSYNC();
    return 0;
}

__device__ void fse_rle(FseView t, uint32_t sym)
{
    WSYNC();
    if (lane_id() == 0) {
        t.e[0] = sym;
        *t.al = 0;
        *t.valid = 1;
    }
    WSYNC();
}

template <int N>
__device__ void load_default(ZLds &L, const int8_t (&def)[N])
{
    WSYNC();
    if (lane_id() < (uint32_t)N) L.norm[lane_id()] = def[lane_id()];
    WSYNC();
}

// Huffman decoding table from L.weights[0..n) (last weight implied), sec. 4.2.1.  Returns 0 / -1.
__device__ int huf_build(ZLds &L, int n)
{
    WSYNC();
    const uint32_t lane = lane_id();
    // lane l keeps the weights of symbols l, 64 + l, 128 + l, 192 + l; per-weight symbol counts come from ballots, so
    // nothing below walks the 256 symbols one LDS read at a time
    uint32_t w[4];
#pragma unroll
    for (int c = 0; c < 4; c++) w[c] = 64 * c + (int)lane < n ? (uint32_t)L.weights[64 * c + lane] : 0u;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; c++) bad = bad || w[c] > 11;
    if (__any(bad)) return -1;
    uint32_t cntw[12];
    uint32_t total = 0;
#pragma unroll
    for (uint32_t wt = 1; wt <= 11; wt++) {
        uint32_t cn = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) cn += (uint32_t)__popcll(__ballot(w[c] == wt));
        cntw[wt] = cn;
        total += cn << (wt - 1);
    }
    if (total == 0) return -1;
    int maxbits = 32 - __clz((int)total);
    if (maxbits > 11) return -1;
    uint32_t rest = (1u << maxbits) - total;
    if (rest & (rest - 1)) return -1;
    const uint32_t lastw = (uint32_t)(32 - __clz((int)rest));  // the implied weight of the last symbol
    if (lane == 0) L.weights[n] = (uint8_t)lastw;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (64 * c + (int)lane == n) w[c] = lastw;
    cntw[lastw] += 1;
    n++;
    if (cntw[1] < 2 || (cntw[1] & 1)) return -1;
    WSYNC();
    // table position of a symbol: everything of lower weight, then the symbols of its own weight in symbol order
    uint32_t basew[12];
    {
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t wt = 1; wt <= 11; wt++) {
            basew[wt] = acc;
            acc += cntw[wt] << (wt - 1);
        }
    }
    uint32_t pos[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t wt = 1; wt <= 11; wt++) {
        uint32_t before = 0;  // symbols of this weight in lower chunks
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint64_t m = __ballot(w[c] == wt);
            if (w[c] == wt) pos[c] = basew[wt] + ((before + (uint32_t)__popcll(m & lanemask_lt())) << (wt - 1));
            before += (uint32_t)__popcll(m);
        }
    }
    // fill, entry by entry (32 per lane): the table holds the symbols by rising weight, a symbol of weight wt in 2^(wt-1) neighbouring
    // entries.  An entry finds its weight class from the classes' first positions, its rank in the class from its distance to that
    // position, and the symbol from the list of symbols in (weight, symbol) order that is written first (over the weights, which are
    // in registers by now).  (Symbol by symbol with the lanes spread over a symbol's entries -- most symbols have one to four -- this
    // was 200 serial steps per table: a twelfth of the kernel.)
    uint32_t firstw[12];  // symbols of lower weights
    {
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t wt = 1; wt <= 11; wt++) {
            firstw[wt] = acc;
            acc += cntw[wt];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (w[c]) L.weights[firstw[w[c]] + ((pos[c] - basew[w[c]]) >> (w[c] - 1u))] = (uint8_t)(64 * c + (int)lane);
    WSYNC();
    const uint32_t size = 1u << maxbits;
    for (uint32_t x = lane; x < size; x += 64) {
        uint32_t wt = 1;
#pragma unroll
        for (uint32_t t = 2; t <= 11; t++) wt += (x >= basew[t]) ? 1u : 0u;  // (classes without symbols share their successor's position)
        uint32_t bw = 0, fw = 0;
#pragma unroll
        for (uint32_t t = 1; t <= 11; t++) {
            bw = wt == t ? basew[t] : bw;
            fw = wt == t ? firstw[t] : fw;
        }
        const uint32_t sym = L.weights[fw + ((x - bw) >> (wt - 1u))];
        L.huf[x] = (uint16_t)(sym | (((uint32_t)maxbits + 1u - wt) << 8));
    }
    if (lane == 0) {
        L.huf_bits = (uint32_t)maxbits;
        L.huf_valid = 1;
    }
    WSYNC();
    return 0;
}

// Huffman tree description at absolute byte p0 (<= n bytes): returns bytes consumed or -1.
__device__ int huf_read(ZLds &L, const Bits &b, uint32_t p0, uint32_t n ZSTAT_PARAM)
{
    if (n < 1) return -1;
    const uint32_t hb = byte_at(b, p0);
    int nw = 0;
    uint32_t used;
    WSYNC();
    if (hb >= 128) {
        nw = (int)hb - 127;
        used = 1 + (uint32_t)(nw + 1) / 2;
        if (used > n) return -1;
        for (int i = lane_id(); i < nw; i += 64) {
            uint32_t v = byte_at(b, p0 + 1 + (uint32_t)i / 2);
            L.weights[i] = (uint8_t)((i & 1) ? (v & 15u) : (v >> 4));
        }
    } else {
        used = 1 + hb;
        if (hb == 0 || used > n) return -1;
        int al, nsym;
        ZT_BEGIN(zt15);
        int c = fse_read_ncount(L, b, p0 + 1, hb, 6, 12, al, nsym);
        if (c < 0) return -1;
        if (fse_build(L, view(L.wt), nsym, al)) return -1;
        ZT_END(15, zt15);
        ZT_BEGIN(zt16);
        BackBits s;
        if (!bb_init(b, s, p0 + 1 + (uint32_t)c, hb - (uint32_t)c)) return -1;
        // the weight stream is at most 127 bytes: lane i keeps its i-th dword, so the (serial, two-state) FSE decode below
        // reads bits with readlane instead of a memory round trip per symbol
        const uint32_t d0 = rdfirst(s.lo >> 5), lo0 = rdfirst(s.lo);
        const uint32_t mydw = d0 + lane_id() < b.total_dw ? b.g32[d0 + lane_id()] : 0u;
        int32_t avail = (int32_t)rdfirst((uint32_t)s.avail);
        // Everything in the loop below is the same in every lane and is written so that it stays on the scalar unit: the stream's
        // dwords and the table's entries are read with readlane, the shifts are 64-bit scalar shifts, every lane stores the (same)
        // weight.  (With a vector funnel shift in it the compiler kept the states in vector registers and paid a readfirstlane, an
        // exec-mask region and hazard no-ops per weight: 84 instructions per weight on a serial chain.)
        auto take = [&](uint32_t nbits) -> uint32_t {  // next nbits (<= 24) of the stream; bits below its start read as zero
            uint32_t v = 0;
            if (nbits != 0 && avail > 0) {
                const uint32_t have = (uint32_t)avail >= nbits ? nbits : (uint32_t)avail;
                const uint32_t pos = lo0 + (uint32_t)avail - have;
                const uint32_t i = ((pos >> 5) - d0) & 63u;
                const uint64_t ww = ((uint64_t)rdlane(mydw, (i + 1u) & 63u) << 32) | rdlane(mydw, i);
                v = ((uint32_t)(ww >> (pos & 31u)) & ((1u << have) - 1u)) << (nbits - have);
            }
            avail -= (int32_t)nbits;
            return v;
        };
        const uint32_t al_u = rdfirst((uint32_t)al);
        uint32_t sa = take(al_u), sb = take(al_u);
        if (avail < 0) return -1;
        // the weights' table has at most 64 states: lane i keeps entry i, a state's entry is a readlane away
        const uint32_t mye = lane_id() < (1u << al_u) ? L.wt.e[lane_id()] : 0u;
        for (;;) {  // the two states take turns: sa decodes, sb is the other one
            if (nw > 253) return -1;
            const uint32_t e = rdlane(mye, sa & 63u);
            L.weights[nw] = (uint8_t)(e & 63u);
            nw++;
            const uint32_t nxt = (e >> 16) + take((e >> 6) & 15u);
            if (avail < 0) {
                L.weights[nw] = (uint8_t)(rdlane(mye, sb & 63u) & 63u);
                nw++;
                break;
            }
            sa = sb;
            sb = nxt;
        }
        ZT_END(16, zt16);
    }
    if (nw > 255) return -1;
    ZT_BEGIN(zt17);
    if (huf_build(L, nw)) return -1;
    ZT_END(17, zt17);
    return (int)used;
}

// XXH64 (seed 0), lanes 0..3 own the four accumulators.  Three steps so that the streaming decoder can carry the state
// from block to block: xxh_init(), xxh_stripes() over whole 32-byte stripes, xxh_finish() with the bytes behind the last stripe.
// `stage`: 4 * stage_stripes qwords of LDS scratch.
__device__ __forceinline__ uint64_t xxh_init()
{
    constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL;
    const uint32_t lane = lane_id();
    return lane == 0 ? P1 + P2 : lane == 1 ? P2 : lane == 2 ? 0ULL : 0ULL - P1;
}

struct XxhView {  // aligned dword view of a byte range; reads stay inside dwords that hold bytes of it
    const uint32_t *p32;
    uint32_t pmis;
    __device__ __forceinline__ explicit XxhView(const uint8_t *p) : p32((const uint32_t *)(p - ((uintptr_t)p & 3u))), pmis((uint32_t)((uintptr_t)p & 3u)) {}
    __device__ __forceinline__ uint64_t rd64(uint32_t off) const
    {
        const uint32_t i = (pmis + off) >> 2, sh = ((pmis + off) & 3u) * 8u;
        const uint32_t d0 = p32[i], d1 = p32[i + 1], d2 = sh ? p32[i + 2] : 0u;
        return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
    }
};

// (out of line; the pointers carry their address spaces, else every access here is a flat one that counts on both wait counters)
__device__ void xxh_stripes(uint64_t *stage_, const uint32_t stage_stripes, const uint8_t *p_, uint32_t stripes, uint64_t &acc)
{
    LDS_AS uint64_t *const stage = (LDS_AS uint64_t *)stage_;
    const GAS uint8_t *const p = (const GAS uint8_t *)p_;
    constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    const uint32_t lane = lane_id();
    // The accumulator recurrence acc = rotl(acc + in * P2, 31) * P1 is serial per 8-byte column, but the products
    // in * P2 are not: all 64 lanes form them for up to 128 stripes at a time into `stage` (4 KB of LDS), then lanes 0..3
    // run their columns over the staged products -- one multiplication per dependent step instead of two, and no
    // address arithmetic or loads on the chain.
    for (uint32_t s0 = 0; s0 < stripes; s0 += stage_stripes) {
        const uint32_t ns = stripes - s0 < stage_stripes ? stripes - s0 : stage_stripes;
        WSYNC();
        // one 8-byte load at any alignment per product (gfx950 takes them), all eight of a lane requested before the first is
        // used: no branch around a load (a lane past the batch's end re-reads its last qword and stores nothing)
        struct __attribute__((packed, aligned(1))) U64u {
            uint64_t v;
        };
        uint64_t in[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t q = 64u * k + lane;  // qword of the batch: stripe q / 4, column q % 4
            const uint32_t qq = q < 4u * ns ? q : 4u * ns - 1u;
            in[k] = ((const GAS U64u *)(p + 32u * s0 + 8u * qq))->v;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t q = 64u * k + lane;
            if (q < 4u * ns) stage[q] = in[k] * P2;
        }
        WSYNC();
        if (lane < 4) {
            uint32_t i = 0;
            for (; i + 8 <= ns; i += 8) {
                uint64_t in[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) in[k] = stage[4u * (i + k) + lane];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) acc = rotl(acc + in[k], 31) * P1;
            }
            for (; i < ns; i++) acc = rotl(acc + stage[4u * i + lane], 31) * P1;
        }
    }
    WSYNC();
}

// total: bytes hashed altogether (stripes and tail); tail[0..tail_n): the bytes behind the last whole stripe (< 32).  Low 32 bits.
__device__ uint32_t xxh_finish(uint64_t acc, uint64_t total, const uint8_t *tail, uint32_t tail_n)
{
    constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                       P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto round1 = [&](uint64_t a, uint64_t in) { return rotl(a + in * P2, 31) * P1; };
    const XxhView V(tail);
    uint64_t h;
    if (total >= 32) {
        uint64_t v[4];
        for (int k = 0; k < 4; k++) {
            uint32_t lo = rdlane((uint32_t)acc, (uint32_t)k), hi = rdlane((uint32_t)(acc >> 32), (uint32_t)k);
            v[k] = ((uint64_t)hi << 32) | lo;
        }
        h = rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18);
        for (int k = 0; k < 4; k++) h = (h ^ round1(0, v[k])) * P1 + P4;
    } else {
        h = P5;
    }
    h += total;
    uint32_t off = 0;
    while (off + 8 <= tail_n) {
        h ^= round1(0, V.rd64(off));
        h = rotl(h, 27) * P1 + P4;
        off += 8;
    }
    if (off + 4 <= tail_n) {
        uint64_t v = 0;
        for (int k = 3; k >= 0; k--) v = (v << 8) | tail[off + (uint32_t)k];
        h ^= v * P1;
        h = rotl(h, 23) * P2 + P3;
        off += 4;
    }
    while (off < tail_n) {
        h ^= (uint64_t)tail[off] * P5;
        h = rotl(h, 11) * P1;
        off++;
    }
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return (uint32_t)h;
}

// XXH64 (seed 0) of p[0..n) in one go.  `stage`: 512 qwords of LDS scratch (anything that is dead once the frame's blocks are done).
__device__ uint32_t wave_xxh64_low32(uint64_t *stage, const uint8_t *p, uint32_t n)
{
    uint64_t acc = xxh_init();
    const uint32_t stripes = n >> 5;
    if (stripes) xxh_stripes(stage, 128, p, stripes, acc);
    return xxh_finish(acc, n, p + (stripes << 5), n & 31u);
}

// wave-cooperative forward copy of n bytes; dst and src may overlap with dst - src = period >= 1
// (the LZ77 replicate case) or be disjoint
// forward copy of n bytes (dst below src when they overlap): 16 bytes per lane and trip at any alignment, 1 KB per trip
struct __attribute__((packed, aligned(1))) ZU128 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ void wave_copy_bytes(uint8_t *dst, const uint8_t *src, uint32_t n)
{
    const uint32_t lane = lane_id(), whole = n & ~15u;
    for (uint32_t j = 16u * lane; j < whole; j += 1024u) {
        const ZU128 v = *(const ZU128 *)(src + j);
        *(ZU128 *)(dst + j) = v;
    }
    if (whole + lane < n) dst[whole + lane] = src[whole + lane];
}

__device__ void wave_match_copy(uint8_t *dst, uint32_t offset, uint32_t n)
{
    const uint32_t lane = lane_id();
    const uint8_t *src = dst - offset;
    if (offset >= n) {
        for (uint32_t j = lane; j < n; j += 64) dst[j] = src[j];
    } else if (offset >= 64) {
        // chunks of 64 bytes never read what the same chunk writes
        for (uint32_t base = 0; base < n; base += 64) {
            uint32_t j = base + lane;
            uint8_t v = j < n ? src[j] : (uint8_t)0;
            if (j < n) dst[j] = v;
        }
    } else {
        // short period: every byte is a copy of one of the `offset` bytes before dst
        for (uint32_t j = lane; j < n; j += 64) dst[j] = src[j % offset];
    }
}

#ifndef CHIP_ZSTD_WAVES
#define CHIP_ZSTD_WAVES 3  // waves per SIMD the register budget is set for (157 VGPRs); at 4 the LDS (11.2 KB) would allow 14 waves per CU, but 34 registers spill: 5.15 against 4.98 ms
#endif
__global__ __launch_bounds__(64, CHIP_ZSTD_WAVES) void zstd_kernel(BatchArgs a, int wlog_max)
{
    __shared__ ZLds L;
    if (blockIdx.x >= (a.sel_n ? *a.sel_n : a.n)) return;
    const uint32_t u = a.sel ? a.sel[blockIdx.x] : blockIdx.x;
    const uint32_t lane = lane_id();
    const uint8_t *gin = a.in_base + a.in_off[u];
    const uint32_t in_len = a.in_len[u];
    uint8_t *gout = a.out_base + a.out_off[u];
    const uint32_t cap = a.out_cap[u];
#ifdef CHIP_STATS
    unsigned long long zst_[24] = {};
    unsigned long long *const zst = zst_;
    const unsigned long long zt0 = __builtin_readcyclecounter();
#endif

    Bits b;
    const uint32_t mis = (uint32_t)((uintptr_t)gin & 3u);
    b.g32 = (const uint32_t *)(gin - mis);
    b.total_dw = (mis + in_len + 3u) >> 2;
    const uint32_t B0 = mis;  // absolute byte index of the unit's first byte
    const uint32_t END = mis + in_len;

    int32_t status = ST_RUNNING;
    uint32_t ip = B0;  // absolute byte cursor
    uint32_t opos = 0;  // output cursor
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    bool has_checksum = false, has_fcs = false;
    uint64_t fcs = 0, window = 0, out_limit = ~0ULL;
    if (lane == 0) {
        L.huf_valid = 0;
        L.ll.valid = L.of.valid = L.ml.valid = 0;
    }
    if (lane < 20) L.lltab[lane] = LL_BASE[16 + lane] | ((uint32_t)LL_BITS[16 + lane] << 24);
    if (lane < 21) L.mltab[lane] = ML_BASE[32 + lane] | ((uint32_t)ML_BITS[32 + lane] << 24);
    WSYNC();

#define ZFAIL(code) do { status = -(code); goto done; } while (0)
#define ZNEED_INPUT() do { status = CHIP_NEED_INPUT; goto done; } while (0)

    uint32_t *const rs = a.resume;  // streaming decoder: checkpoint blob of this (single) unit, else nullptr
    bool resumed = false, blocks_done = false;
    // output bytes of this frame that the host has dropped in front of gout (streaming only): positions compared with the
    // frame's content size or output limit count them, offsets can only reach what is still there
    const uint64_t dropped = rs ? (uint64_t)rs[12] | ((uint64_t)rs[13] << 32) : 0ull;
    if (rs && rs[0] != 0 && rs[0] - 1u <= in_len && rs[1] <= cap) {
        resumed = true;
        ip = B0 + rs[0] - 1u;
        opos = rs[1];
        rep0 = rs[2];
        rep1 = rs[3];
        rep2 = rs[4];
        has_checksum = rs[5] & 1u;
        has_fcs = (rs[5] >> 1) & 1u;
        blocks_done = (rs[5] >> 2) & 1u;
        fcs = rs[6] | ((uint64_t)rs[7] << 32);
        window = rs[8] | ((uint64_t)rs[9] << 32);
        out_limit = rs[10] | ((uint64_t)rs[11] << 32);
        for (uint32_t k = lane; k < ZSAVE_WORDS; k += 64) ((uint32_t *)&L)[k] = rs[ZRES_HDR + k];
        WSYNC();
    }
    if (rs && !resumed) {  // the running checksum starts with the frame
        const uint64_t acc0 = xxh_init();
        if (lane < 4) {
            rs[16 + 2 * lane] = (uint32_t)acc0;
            rs[17 + 2 * lane] = (uint32_t)(acc0 >> 32);
        }
        if (lane == 0) rs[14] = rs[15] = 0;
    }
    // streaming: the running XXH64 is brought up to the last whole 32-byte stripe of the output (the host drops nothing behind it)
    auto hash_upto_opos = [&](uint64_t *stage, uint32_t stage_stripes, uint64_t &acc, uint32_t &rel) {
        const uint64_t hashed = (uint64_t)rs[14] | ((uint64_t)rs[15] << 32);
        acc = lane < 4 ? (uint64_t)rs[16 + 2 * lane] | ((uint64_t)rs[17 + 2 * lane] << 32) : 0ull;
        rel = (uint32_t)(hashed - dropped);
        const uint32_t ns = (opos - rel) >> 5;
        if (ns) xxh_stripes(stage, stage_stripes, gout + rel, ns, acc);
        rel += ns << 5;
        const uint64_t h2 = dropped + rel;
        if (lane < 4) {
            rs[16 + 2 * lane] = (uint32_t)acc;
            rs[17 + 2 * lane] = (uint32_t)(acc >> 32);
        }
        if (lane == 0) {
            rs[14] = (uint32_t)h2;
            rs[15] = (uint32_t)(h2 >> 32);
        }
    };
    auto save_checkpoint = [&](bool all_done) {
        if (!rs) return;
        WSYNC();
        if (has_checksum) {  // (between blocks the sequence window / copy parameters are dead: 48 stripes of staging)
            uint64_t acc;
            uint32_t rel;
            hash_upto_opos((uint64_t *)&L.seqwin[(offsetof(ZLds, seqwin) & 4u) ? 1 : 0], 48, acc, rel);  // (qword aligned)
        }
        for (uint32_t k = lane; k < ZSAVE_WORDS; k += 64) rs[ZRES_HDR + k] = ((const uint32_t *)&L)[k];
        if (lane == 0) {
            rs[1] = opos;
            rs[2] = rep0;
            rs[3] = rep1;
            rs[4] = rep2;
            rs[5] = (has_checksum ? 1u : 0u) | (has_fcs ? 2u : 0u) | (all_done ? 4u : 0u);
            rs[6] = (uint32_t)fcs;
            rs[7] = (uint32_t)(fcs >> 32);
            rs[8] = (uint32_t)window;
            rs[9] = (uint32_t)(window >> 32);
            rs[10] = (uint32_t)out_limit;
            rs[11] = (uint32_t)(out_limit >> 32);
            rs[0] = 1u + (ip - B0);
        }
    };

    // ---- frame header (sec. 3.1.1.1) ------------------------------------------------------------
    if (!resumed) {
        if (END - ip < 4) ZNEED_INPUT();
        uint32_t magic = rd32_at(b, ip * 8u);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame (sec. 3.1.2): a frame without content
            if (END - ip < 8) ZNEED_INPUT();
            uint32_t sz = rd32_at(b, (ip + 4) * 8u);
            if ((uint64_t)(END - ip) < 8ull + sz) ZNEED_INPUT();
            ip += 8 + sz;
            status = CHIP_FINISHED;
            goto done;
        }
        if (magic != 0xFD2FB528u) ZFAIL(ZSTD_E_PREFIX_UNKNOWN);
        if (END - ip < 5) ZNEED_INPUT();
        const uint32_t fhd = byte_at(b, ip + 4), fcs_flag = fhd >> 6, single = (fhd >> 5) & 1u, did_flag = fhd & 3u;
        const uint32_t did_sz = did_flag == 3 ? 4u : did_flag, fsz = fcs_flag ? (1u << fcs_flag) : single;
        const uint32_t hsz = 5 + (single ? 0u : 1u) + did_sz + fsz;
        if (END - ip < hsz) ZNEED_INPUT();
        if (fhd & 0x08u) ZFAIL(ZSTD_E_FRAMEPARAM_UNSUPPORTED);
        uint32_t q = ip + 5;
        if (!single) {
            uint32_t wd = byte_at(b, q++);
            uint32_t wlog = (wd >> 3) + 10;
            if (wlog > 31) ZFAIL(ZSTD_E_WINDOW_TOO_LARGE);
            window = 1ull << wlog;
            window += (window >> 3) * (wd & 7u);
        }
        uint32_t did = 0;
        for (uint32_t k = 0; k < did_sz; k++) did |= byte_at(b, q++) << (8 * k);
        for (uint32_t k = 0; k < fsz; k++) fcs |= (uint64_t)byte_at(b, q++) << (8 * k);
        if (fcs_flag == 1) fcs += 256;
        has_fcs = fsz > 0;
        if (single) window = fcs;
        if (window > (1ull << wlog_max)) ZFAIL(ZSTD_E_WINDOW_TOO_LARGE);
        if (did != 0) ZFAIL(ZSTD_E_DICT_WRONG);
        has_checksum = (fhd >> 2) & 1u;
        {
            uint64_t bm = window < BLOCK_MAX ? window : BLOCK_MAX, ring = window + bm + 64;
            if (has_fcs && fcs < ring) out_limit = fcs;  // libzstd's output buffer is no larger than the content size
        }
        ip += hsz;
    }

    // ---- blocks (sec. 3.1.1.2) --------------------------------------------------------------------
    while (!blocks_done) {
        if (END - ip < 3) ZNEED_INPUT();
        const uint32_t bh = rd32_at(b, ip * 8u) & 0xffffffu;
        const uint32_t last = bh & 1u, type = (bh >> 1) & 3u, bsz = bh >> 3;
        const uint32_t bmax = window < BLOCK_MAX ? (uint32_t)window : BLOCK_MAX;
        if (type == 3) ZFAIL(ZSTD_E_CORRUPTION);
        if (bsz > bmax) ZFAIL(ZSTD_E_CORRUPTION);
        if (type == 0) {  // raw block: streams through as far as the input goes
            if (dropped + opos + bsz > out_limit) ZFAIL(70);
            ip += 3;
            uint32_t avail = END - ip, k = bsz < avail ? bsz : avail;
            if (k > cap - opos) {
                status = CHIP_NEED_OUTPUT;
                goto done;
            }
            wave_copy_bytes(gout + opos, (const uint8_t *)b.g32 + ip, k);
            opos += k;
            ip += k;
            if (k < bsz) ZNEED_INPUT();
        } else if (type == 1) {
            if (END - ip < 4) ZNEED_INPUT();
            if (dropped + opos + bsz > out_limit) ZFAIL(70);
            if (bsz > cap - opos) {
                status = CHIP_NEED_OUTPUT;
                goto done;
            }
            const uint8_t v = (uint8_t)byte_at(b, ip + 3);
            for (uint32_t j = lane; j < bsz; j += 64) gout[opos + j] = v;
            opos += bsz;
            ip += 4;
        } else {
            if (END - ip < 3 + bsz) ZNEED_INPUT();
            // ---- compressed block ----------------------------------------------------------------
            const uint32_t bp = ip + 3, bend = bp + bsz;
            if (bsz < 1) ZFAIL(ZSTD_E_CORRUPTION);
            const uint32_t b0 = byte_at(b, bp), ltype = b0 & 3u, sf = (b0 >> 2) & 3u;
            uint32_t hl, regen, comp = 0, streams = 1;
            if (ltype < 2) {
                if (sf == 0 || sf == 2) {
                    hl = 1;
                    regen = b0 >> 3;
                } else if (sf == 1) {
                    if (bsz < 2) ZFAIL(ZSTD_E_CORRUPTION);
                    hl = 2;
                    regen = (b0 >> 4) | (byte_at(b, bp + 1) << 4);
                } else {
                    if (bsz < 3) ZFAIL(ZSTD_E_CORRUPTION);
                    hl = 3;
                    regen = (b0 >> 4) | (byte_at(b, bp + 1) << 4) | (byte_at(b, bp + 2) << 12);
                }
            } else {
                if (bsz < 5) ZFAIL(ZSTD_E_CORRUPTION);
                const uint64_t v = (uint64_t)rd32_at(b, bp * 8u) | ((uint64_t)byte_at(b, bp + 4) << 32);
                if (sf == 0 || sf == 1) {
                    hl = 3;
                    regen = (uint32_t)(v >> 4) & 0x3ffu;
                    comp = (uint32_t)(v >> 14) & 0x3ffu;
                    streams = sf ? 4 : 1;
                } else if (sf == 2) {
                    hl = 4;
                    regen = (uint32_t)(v >> 4) & 0x3fffu;
                    comp = (uint32_t)(v >> 18) & 0x3fffu;
                    streams = 4;
                } else {
                    hl = 5;
                    regen = (uint32_t)(v >> 4) & 0x3ffffu;
                    comp = (uint32_t)(v >> 22) & 0x3ffffu;
                    streams = 4;
                }
            }
            if (regen > BLOCK_MAX) ZFAIL(ZSTD_E_CORRUPTION);
            uint32_t p = bp + hl, left = bsz - hl;
            // literal source: 0 = bytes in the input at lit_in, 1 = one repeated byte, 2 = parked at gout+lit_out
            uint32_t lit_mode = 0, lit_in = 0, lit_out = 0, lit_rle = 0;
            if (ltype == 0) {
                if (regen > left) ZFAIL(ZSTD_E_CORRUPTION);
                lit_in = p;
                p += regen;
                left -= regen;
            } else if (ltype == 1) {
                if (left < 1) ZFAIL(ZSTD_E_CORRUPTION);
                lit_mode = 1;
                lit_rle = byte_at(b, p);
                p += 1;
                left -= 1;
            } else {
                if (comp > left) ZFAIL(ZSTD_E_CORRUPTION);
                uint32_t lp = p, lleft = comp;
                if (ltype == 2) {
                    ZT_BEGIN(zt13);
                    int c = huf_read(L, b, lp, lleft ZSTAT_ARG);
                    ZT_END(13, zt13);
                    if (c < 0) ZFAIL(ZSTD_E_CORRUPTION);
                    lp += (uint32_t)c;
                    lleft -= (uint32_t)c;
                } else if (!L.huf_valid) ZFAIL(30);  // treeless without a previous table: dictionary_corrupted
                if (regen > cap - opos) {
                    status = CHIP_NEED_OUTPUT;
                    goto done;
                }
                lit_mode = 2;
                lit_out = cap - regen;
                uint32_t sz[4] = {lleft, 0, 0, 0}, cnt_[4] = {regen, 0, 0, 0};
                uint32_t st0 = lp;
                if (streams == 4) {
                    if (lleft < 10) ZFAIL(ZSTD_E_CORRUPTION);
                    uint32_t j0 = rd32_at(b, lp * 8u), j1 = rd32_at(b, (lp + 4) * 8u);
                    sz[0] = j0 & 0xffffu;
                    sz[1] = j0 >> 16;
                    sz[2] = j1 & 0xffffu;
                    if (6 + sz[0] + sz[1] + sz[2] > lleft) ZFAIL(ZSTD_E_CORRUPTION);
                    sz[3] = lleft - 6 - sz[0] - sz[1] - sz[2];
                    const uint32_t seg = (regen + 3) / 4;
                    if (seg * 3 > regen) ZFAIL(ZSTD_E_CORRUPTION);
                    cnt_[0] = cnt_[1] = cnt_[2] = seg;
                    cnt_[3] = regen - 3 * seg;
                    st0 = lp + 6;
                }
                bool sbad = false;
                if (streams == 4) {
                    // 16 lanes per stream (see huf_decode4)
                    const uint32_t g = lane >> 4;
                    uint32_t myoff = st0, myout = lit_out;
                    for (uint32_t k = 0; k < g; k++) {
                        myoff += sz[k];
                        myout += cnt_[k];
                    }
                    BackBits s;
                    const bool okb = bb_init(b, s, myoff, sz[g]);
                    if (__any(!okb)) ZFAIL(ZSTD_E_CORRUPTION);
                    HufStream hs;
                    hs.lo = s.lo;
                    hs.top = s.lo + (uint32_t)s.avail;
                    hs.want = cnt_[g];
                    hs.done = 0;
                    hs.out = myout;
                    WSYNC();
#ifndef CHIP_EXP_NOLIT  // (ablation: no literal decoding at all)
                    ZT_BEGIN(zt1);
                    // the walk's scratch rows: the free part of the frame's own output range, if it is large enough (see huf_decode4)
                    uint32_t *scr = nullptr;
                    {
                        // (pointer arithmetic on gout, not an integer round trip: the accesses stay global_*, a flat store would also
                        // count on the LDS counter and every LDS wait of the walk would wait for it)
                        const uint32_t pad = (4u - (uint32_t)((uintptr_t)(gout + opos) & 3u)) & 3u;
                        if ((uint64_t)opos + pad + HSCR_BYTES <= lit_out) scr = (uint32_t *)(gout + opos + pad);
                    }
                    const bool huf_ok = huf_decode4(L, b, hs, gout, scr ZSTAT_ARG);
                    ZT_END(1, zt1);
                    if (!huf_ok) ZFAIL(ZSTD_E_CORRUPTION);
#endif
                    WSYNC();
                } else if (lane < streams) {  // a single stream (small literal sections): one lane, symbol by symbol
                    uint32_t myoff = st0, myout = lit_out;
                    for (uint32_t k = 0; k < lane; k++) {
                        myoff += sz[k];
                        myout += cnt_[k];
                    }
                    BackBits s;
                    if (!bb_init(b, s, myoff, sz[lane])) sbad = true;
                    else {
                        // per-lane register bit buffer (next bit at bit 63) fed by aligned dword loads issued one
                        // refill ahead, so the table lookup is the only latency on the symbol chain
                        const int32_t lo = (int32_t)s.lo;
                        auto load_dw = [&](int32_t bit) -> uint32_t {
                            const int32_t i = bit >> 5;
                            if (bit + 32 <= lo || i < 0 || (uint32_t)i >= b.total_dw) return 0u;
                            uint32_t dw = b.g32[i];
                            if (bit < lo) dw &= ~((1u << (lo - bit)) - 1u);
                            return dw;
                        };
                        const uint32_t top = s.lo + (uint32_t)s.avail;
                        int32_t ptr = (int32_t)(top & ~31u), cnt = (int32_t)(top & 31u);
                        uint64_t buf = cnt ? (uint64_t)(load_dw(ptr) & ((1u << cnt) - 1u)) << (64 - cnt) : 0ull;
                        uint32_t nextdw = load_dw(ptr - 32);
                        int32_t avail = s.avail;
                        const uint32_t hbits = L.huf_bits;
#ifdef CHIP_EXP_NOHUF
                        const uint32_t ncnt = 0;
                        avail = 0;
#else
                        const uint32_t ncnt = cnt_[lane];
#endif
                        for (uint32_t i = 0; i < ncnt; i++) {
                            if (cnt <= 32) {
                                ptr -= 32;
                                buf |= (uint64_t)nextdw << (32 - cnt);
                                cnt += 32;
                                nextdw = load_dw(ptr - 32);
                            }
                            const uint32_t e = L.huf[(uint32_t)(buf >> (64 - hbits))];
                            const uint32_t nb = e >> 8;
                            gout[myout + i] = (uint8_t)e;
                            buf <<= nb;
                            cnt -= (int32_t)nb;
                            avail -= (int32_t)nb;
                            if (avail < 0) {
                                sbad = true;
                                break;
                            }
                        }
                        if (avail != 0) sbad = true;  // the stream must be consumed exactly
                    }
                }
                if (__any(sbad)) ZFAIL(ZSTD_E_CORRUPTION);
                p += comp;
                left -= comp;
            }
            // ---- sequences section (sec. 3.1.1.3.2) ---------------------------------------------
            if (left < 1) ZFAIL(ZSTD_E_CORRUPTION);
            uint32_t nseq = byte_at(b, p);
            if (nseq < 128) {
                p += 1;
                left -= 1;
            } else if (nseq < 255) {
                if (left < 2) ZFAIL(ZSTD_E_CORRUPTION);
                nseq = ((nseq - 128) << 8) + byte_at(b, p + 1);
                p += 2;
                left -= 2;
            } else {
                if (left < 3) ZFAIL(ZSTD_E_CORRUPTION);
                nseq = byte_at(b, p + 1) + (byte_at(b, p + 2) << 8) + 0x7F00u;
                p += 3;
                left -= 3;
            }
            const uint32_t block_out0 = opos;
            uint32_t lpos = 0;
            auto copy_literals = [&](uint32_t n) {
                if (lit_mode == 0) wave_copy_bytes(gout + opos, (const uint8_t *)b.g32 + lit_in + lpos, n);
                else if (lit_mode == 1)
                    for (uint32_t j = lane; j < n; j += 64) gout[opos + j] = (uint8_t)lit_rle;
                else wave_copy_bytes(gout + opos, gout + lit_out + lpos, n);
            };
            if (nseq == 0) {
                if (left != 0) ZFAIL(ZSTD_E_CORRUPTION);
            } else {
                if (left < 1) ZFAIL(ZSTD_E_CORRUPTION);
                const uint32_t modes = byte_at(b, p);
                p += 1;
                left -= 1;
                if (modes & 3u) ZFAIL(ZSTD_E_CORRUPTION);
                ZT_BEGIN(zt14);
                for (int k = 0; k < 3; k++) {
                    const FseView t = k == 0 ? view(L.ll) : k == 1 ? view(L.of) : view(L.ml);
                    const int maxal = k == 1 ? 8 : 9, maxsym = k == 0 ? 35 : k == 1 ? 31 : 52;
                    const uint32_t mode = (modes >> (6 - 2 * k)) & 3u;
                    if (mode == 0) {
                        if (k == 0) load_default(L, LL_DEF);
                        else if (k == 1) load_default(L, OF_DEF);
                        else load_default(L, ML_DEF);
                        if (fse_build(L, t, k == 0 ? 36 : k == 1 ? 29 : 53, k == 1 ? 5 : 6)) ZFAIL(ZSTD_E_CORRUPTION);
                    } else if (mode == 1) {
                        if (left < 1) ZFAIL(ZSTD_E_CORRUPTION);
                        uint32_t sym = byte_at(b, p);
                        if (sym > (uint32_t)maxsym) ZFAIL(ZSTD_E_CORRUPTION);
                        fse_rle(t, sym);
                        p += 1;
                        left -= 1;
                    } else if (mode == 2) {
                        int al, nsym;
                        int c = fse_read_ncount(L, b, p, left, maxal, maxsym, al, nsym);
                        if (c < 0) ZFAIL(ZSTD_E_CORRUPTION);
                        if (fse_build(L, t, nsym, al)) ZFAIL(ZSTD_E_CORRUPTION);
                        p += (uint32_t)c;
                        left -= (uint32_t)c;
                    } else if (!*t.valid) ZFAIL(ZSTD_E_CORRUPTION);
                    if (mode != 3) {
                        // A freshly built table is re-coded for the serial state chain below: [4:0] state bits (used as
                        // they are as v_bfe widths and offsets), [11:5] MINUS (state bits + extra bits of the code) modulo 128
                        // (the sum of the three entries gives minus the bits a sequence consumes: as a signed field it moves
                        // the bit position, and its low six bits are 64 - bits, the shift that brings the state bits of
                        // the 64-bit view to the bottom), [17:12] the code, [31:23] the baseline of the next state.
                        WSYNC();
                        const uint32_t *xt = k == 0 ? L.lltab : L.mltab;
                        const uint32_t xt0 = k == 0 ? 16u : 32u;  // codes below carry no extra bits
                        const uint32_t size = 1u << *t.al;
                        for (uint32_t u = lane; u < size; u += 64) {
                            const uint32_t e = t.e[u], sym = e & 63u, nb = (e >> 6) & 15u, base = e >> 16;
                            const uint32_t xb = k == 1 ? sym : sym < xt0 ? 0u : xt[sym - xt0] >> 24;
                            t.e[u] = nb | (((0u - (nb + xb)) & 127u) << 5) | (sym << 12) | (base << 23);
                        }
                    }
                }
                WSYNC();
                ZT_END(14, zt14);
                BackBits s0;
                if (!bb_init(b, s0, p, left)) ZFAIL(ZSTD_E_CORRUPTION);
                SeqBits s;
                sq_init(L, b, s, s0.lo, s0.avail);
                // the three FSE states, kept as byte offsets into their tables
                uint32_t al = sq_read(L, b, s, L.ll.al) << 2, ao = sq_read(L, b, s, L.of.al) << 2, am = sq_read(L, b, s, L.ml.al) << 2;
                s.avail -= (int32_t)s.used;
                s.used = 0;
                if (s.avail < 0) ZFAIL(ZSTD_E_CORRUPTION);
                const uint8_t *litsrc = lit_mode == 0 ? (const uint8_t *)b.g32 + lit_in : gout + lit_out;
                // sequences are decoded 64 at a time (the FSE state chain is serial; lane j keeps sequence j)
                // and then executed together: prefix sums place every literal run and match, literal bytes
                // and match bytes are copied 256 per step with an owner map (see inflate.hip)
#ifdef CHIP_EXP_NOSEQ
                nseq = 0;
#endif
                for (uint32_t i0 = 0; i0 < nseq; i0 += 64) {
                    ZT_BEGIN(zt4);  // (the whole chunk: what is not chain, phase A or phase B is the parallel part and the placement)
                    const uint32_t cn = nseq - i0 < 64 ? nseq - i0 : 64;
                    uint32_t ll = 0, ml = 0, off = 0;
                    uint32_t dec_bad = 64;  // first sequence of the chunk whose decode is corrupt (verdicts keep stream order)
                    // ---- serial part: only what the next state depends on.  Per sequence: the three state entries,
                    // the number of extra bits they imply (to find the state-update bits) and the three new states.
                    // Lane j keeps sequence j's entries and bit position; the values are cut out in parallel below.
                    // A chunk reads at most 64 x 89 bits, so staging the 8192-bit window once per chunk is enough.
                    s.avail -= (int32_t)s.used;
                    s.used = 0;
                    if ((((int32_t)s.lo + s.avail - 64 - 64 * 96) >> 5) < s.win0) sq_fill(L, b, s);
                    // States are kept as byte offsets into their tables.  Per sequence: six LDS reads in flight together
                    // (the 64 bits below the read position and the three entries), one add of the entries gives the bits
                    // consumed, one 64-bit shift brings the state bits to the bottom, three v_bfe with the entries
                    // themselves as width / offset operands cut them, and lane j keeps the three state offsets of sequence j
                    // to finish it afterwards.  The last sequence of a block updates no state.
                    const uint32_t n_upd = i0 + cn == nseq ? cn - 1u : cn;
                    const int32_t T0 = (int32_t)s.lo + s.avail, Tmin = (int32_t)s.lo;
                    int32_t T = T0;
                    // sequence j's three states are parked in LDS (xpar[3j..]: the copy phase's parameters, dead here) for lane j
                    uint32_t *const sst = L.xpar;
#ifndef CHIP_ZSTD_NO_ASM_CHAIN
                    // (the hand-written loop addresses the tables by their offsets in ZLds: L is the kernel's only LDS object, at 0)
                    bool asm_chain = rdfirst((uint32_t)(uintptr_t)(__attribute__((address_space(3))) ZLds *)&L) == 0u;
#else
                    bool asm_chain = false;
#endif
                    const uint32_t al0 = al, ao0 = ao, am0 = am;
                    uint32_t my_al = 0, my_ao = 0, my_am = 0, my_el = 0, my_eo = 0, my_em = 0, my_tot = 0;
                    for (;;) {
                        LSYNC();  // the copy phase of the chunk before is done with xpar
                        ZT_BEGIN(zt3);
                        ZC(8, 1);
                        __builtin_amdgcn_s_setprio(3);  // a dependent chain: let it go ahead of the other waves' bulk work
                        T = T0;
                        dec_bad = 64;
                        if (asm_chain && n_upd) {
                            // The loop below, written by hand: everything stays in vector registers (the values are the same in
                            // every lane), so the dependent path of a sequence is LDS round trip -> v_add3 -> v_bfe_i32 ->
                            // v_lshrrev_b64 -> v_bfe -> v_lshl_add -> next LDS read, with no detour over the scalar unit.  Every
                            // state is followed at once by the read of its entry, the window words of the next sequence are requested
                            // as soon as its bit position is known.  19 VALU per sequence (the kernel is bound by vector issue).  The
                            // loop has no data-dependent exit: a sequence whose bits do not fit one 64-bit view is seen afterwards by
                            // the lane that finishes it and sends the whole chunk through the C++ loop below; a position below the
                            // stream's start is found in the parallel part.  (v126 / v127: the 64-bit view, a register pair by name.)
                            uint32_t tp = (uint32_t)(T - 64 - 32 * s.win0), j = 0;
                            const uint32_t n_upd_s = rdfirst(n_upd);
                            uint32_t ta, td0, td1, td2, tel, teo, tem, tS, txs, tq, tja = (uint32_t)offsetof(ZLds, xpar);
                            asm volatile(
                                // reads of the first sequence
                                "v_lshrrev_b32 %[a], 3, %[tp]\n\t"
                                "v_and_b32 %[a], 0x3fc, %[a]\n\t"
                                "ds_read_b32 %[d0], %[a] offset:%[OW0]\n\t"
                                "ds_read_b32 %[d1], %[a] offset:%[OW1]\n\t"
                                "ds_read_b32 %[d2], %[a] offset:%[OW2]\n\t"
                                "ds_read_b32 %[el], %[al] offset:%[OL]\n\t"
                                "ds_read_b32 %[em], %[am] offset:%[OM]\n\t"
                                "ds_read_b32 %[eo], %[ao] offset:%[OO]\n\t"
                                "1:\n\t"
                                "ds_write2_b32 %[ja], %[al], %[am] offset1:1\n\t"
                                "ds_write_b32 %[ja], %[ao] offset:8\n\t"
                                "v_add_u32 %[ja], 12, %[ja]\n\t"
                                "s_waitcnt lgkmcnt(2)\n\t"
                                "v_add3_u32 %[S], %[el], %[eo], %[em]\n\t"
                                "v_bfe_i32 %[xs], %[S], 5, 7\n\t"
                                "v_alignbit_b32 v127, %[d2], %[d1], %[tp]\n\t"
                                "v_alignbit_b32 v126, %[d1], %[d0], %[tp]\n\t"
                                "v_lshrrev_b64 v[126:127], %[xs], v[126:127]\n\t"
                                // the next states, each followed at once by the read of its entry
                                "v_add_u32 %[q], %[eo], %[em]\n\t"
                                "v_bfe_u32 %[q], v126, %[q], %[el]\n\t"
                                "v_lshrrev_b32 %[S], 21, %[el]\n\t"
                                "v_lshl_add_u32 %[al], %[q], 2, %[S]\n\t"
                                "ds_read_b32 %[el], %[al] offset:%[OL]\n\t"
                                "v_bfe_u32 %[q], v126, %[eo], %[em]\n\t"
                                "v_lshrrev_b32 %[S], 21, %[em]\n\t"
                                "v_lshl_add_u32 %[am], %[q], 2, %[S]\n\t"
                                "ds_read_b32 %[em], %[am] offset:%[OM]\n\t"
                                "v_bfe_u32 %[q], v126, 0, %[eo]\n\t"
                                "v_lshrrev_b32 %[S], 21, %[eo]\n\t"
                                "v_lshl_add_u32 %[ao], %[q], 2, %[S]\n\t"
                                "ds_read_b32 %[eo], %[ao] offset:%[OO]\n\t"
                                // the bit position and the window words of the next sequence
                                "v_add_u32 %[tp], %[tp], %[xs]\n\t"
                                "v_lshrrev_b32 %[a], 3, %[tp]\n\t"
                                "v_and_b32 %[a], 0x3fc, %[a]\n\t"
                                "ds_read_b32 %[d0], %[a] offset:%[OW0]\n\t"
                                "ds_read_b32 %[d1], %[a] offset:%[OW1]\n\t"
                                "ds_read_b32 %[d2], %[a] offset:%[OW2]\n\t"
                                "s_add_u32 %[j], %[j], 1\n\t"
                                "s_cmp_lt_u32 %[j], %[n]\n\t"
                                "s_cbranch_scc1 1b\n\t"
                                "s_waitcnt lgkmcnt(0)"
                                : [al] "+v"(al), [ao] "+v"(ao), [am] "+v"(am), [tp] "+v"(tp), [ja] "+v"(tja), [j] "+s"(j), [a] "=&v"(ta), [d0] "=&v"(td0),
                                  [d1] "=&v"(td1), [d2] "=&v"(td2), [el] "=&v"(tel), [eo] "=&v"(teo), [em] "=&v"(tem), [S] "=&v"(tS), [xs] "=&v"(txs),
                                  [q] "=&v"(tq)
                                : [n] "s"(n_upd_s), [OL] "n"(offsetof(ZLds, ll)), [OO] "n"(offsetof(ZLds, of)), [OM] "n"(offsetof(ZLds, ml)),
                                  [OW0] "n"(offsetof(ZLds, seqwin)), [OW1] "n"(offsetof(ZLds, seqwin) + 4), [OW2] "n"(offsetof(ZLds, seqwin) + 8)
                                : "v126", "v127", "scc", "memory");
                            T = (int32_t)tp + 64 + 32 * s.win0;
                        } else {
                            for (uint32_t j = 0; j < n_upd; j++) {
                                int32_t wi = ((T - 64) >> 5) - s.win0;
                                wi = wi < 0 ? 0 : wi;
                                const uint32_t d0 = L.seqwin[wi], d1 = L.seqwin[wi + 1], d2 = L.seqwin[wi + 2];
                                const uint32_t el = *(const uint32_t *)((const char *)L.ll.e + al), eo = *(const uint32_t *)((const char *)L.of.e + ao),
                                               em = *(const uint32_t *)((const char *)L.ml.e + am);
                                if (lane == 0) {  // lane j finishes sequence j
                                    sst[3 * j] = al;
                                    sst[3 * j + 1] = am;
                                    sst[3 * j + 2] = ao;
                                }
                                const uint32_t S = el + eo + em;  // [4:0] the three state-bit counts, [11:5] minus (those + the extra bits)
                                const uint32_t tot = (0u - (S >> 5)) & 127u;
                                uint32_t whi = __builtin_amdgcn_alignbit(d2, d1, (uint32_t)T), wlo = __builtin_amdgcn_alignbit(d1, d0, (uint32_t)T);
                                uint32_t x = 64u - tot;
                                if (tot > 64u) {
                                    // rare: extras and state bits do not fit one 64-bit view; take a second one below the extras
                                    const uint32_t n3 = S & 31u;
                                    const int32_t q2 = T - (int32_t)(tot - n3) - 64;
                                    int32_t w2 = (q2 >> 5) - s.win0;
                                    w2 = w2 < 0 ? 0 : w2;
                                    const uint32_t e0 = L.seqwin[w2], e1 = L.seqwin[w2 + 1], e2 = L.seqwin[w2 + 2];
                                    whi = __builtin_amdgcn_alignbit(e2, e1, (uint32_t)q2);
                                    wlo = __builtin_amdgcn_alignbit(e1, e0, (uint32_t)q2);
                                    x = 64u - n3;
                                }
                                const uint32_t W = (uint32_t)((((uint64_t)whi << 32) | wlo) >> (x & 63u));  // LL, ML, OF state bits, OF lowest
                                // next state = baseline + bits; entry bits [22:18] are zero, so entry >> 21 is the baseline times four
                                al = lshl2_add(__builtin_amdgcn_ubfe(W, eo + em, el), el >> 21);
                                am = lshl2_add(__builtin_amdgcn_ubfe(W, eo, em), em >> 21);
                                ao = lshl2_add(__builtin_amdgcn_ubfe(W, 0, eo), eo >> 21);
                                T -= (int32_t)tot;
                                if (T < Tmin) {
                                    dec_bad = j;
                                    break;
                                }
                            }
                        }
                        if (dec_bad == 64 && n_upd < cn) {  // the block's last sequence: extras only, no state update
                            const uint32_t S = *(const uint32_t *)((const char *)L.ll.e + al) + *(const uint32_t *)((const char *)L.of.e + ao) +
                                               *(const uint32_t *)((const char *)L.ml.e + am);
                            if (lane == 0) {
                                sst[3 * n_upd] = al;
                                sst[3 * n_upd + 1] = am;
                                sst[3 * n_upd + 2] = ao;
                            }
                            T -= (int32_t)(((0u - (S >> 5)) & 127u) - (S & 31u));
                            if (T < Tmin) dec_bad = n_upd;
                        }
                        __builtin_amdgcn_s_setprio(0);
                        ZT_END(3, zt3);
                        LSYNC();
                        // lane j takes sequence j's states and reads its entries again
                        const bool have = lane < cn && lane < dec_bad;
                        my_el = my_eo = my_em = my_tot = 0;
                        uint32_t full_tot = 0;
                        if (have) {
                            my_al = sst[3 * lane];
                            my_am = sst[3 * lane + 1];
                            my_ao = sst[3 * lane + 2];
                            my_el = *(const uint32_t *)((const char *)L.ll.e + my_al);
                            my_eo = *(const uint32_t *)((const char *)L.of.e + my_ao);
                            my_em = *(const uint32_t *)((const char *)L.ml.e + my_am);
                            const uint32_t S = my_el + my_eo + my_em;
                            full_tot = my_tot = (0u - (S >> 5)) & 127u;
                            if (i0 + lane + 1 == nseq) my_tot -= S & 31u;
                        }
#ifdef CHIP_EXP_FORCE_FALLBACK  // test hook: every chunk takes the way back through the C++ loop
                        if (asm_chain) {
#else
                        if (asm_chain && __any(lane < n_upd && full_tot > 64u)) {
#endif
                            // rare: some sequence's bits did not fit one 64-bit view -- the states behind it are wrong: the chunk is
                            // done again by the C++ loop
                            al = al0;
                            ao = ao0;
                            am = am0;
                            asm_chain = false;
                            continue;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_setprio(0);
                    s.avail = T - (int32_t)s.lo;
#ifdef CHIP_EXP_CHAINONLY
                    if (my_al != 0xffffffffu) continue;
#endif
                    // ---- parallel part: lane j reads sequence j's entries again, finds its bit position by a prefix sum of
                    // the bits consumed, cuts the extra bits from the window and forms the values
                    uint32_t ov = 4;
                    {
                        const bool have = lane < cn && lane < dec_bad;
                        const uint32_t tot_incl = wave_incl_scan(my_tot);
                        const int32_t my_top = T0 - (int32_t)(tot_incl - my_tot);
                        // the first sequence that reads below the stream's start is the corrupt one (the chain itself ran on)
                        const uint64_t overm = __ballot(have && T0 - (int32_t)tot_incl < Tmin);
                        if (overm) {
                            const uint32_t fo = (uint32_t)__ffsll((long long)overm) - 1u;
                            dec_bad = fo < dec_bad ? fo : dec_bad;
                        }
                        const bool mine = have && lane < dec_bad;
                        if (mine) {
                            const uint32_t oc = (my_eo >> 12) & 63u, mc = (my_em >> 12) & 63u, lc = (my_el >> 12) & 63u;
                            const uint32_t xl = ((0u - (my_el >> 5)) & 127u) - (my_el & 31u), xm = ((0u - (my_em >> 5)) & 127u) - (my_em & 31u);
                            const int32_t q = my_top - 64;
                            int32_t wi = (q >> 5) - s.win0;
                            wi = wi < 0 ? 0 : (wi > 253 ? 253 : wi);
                            const uint32_t d0 = L.seqwin[wi], d1 = L.seqwin[wi + 1], d2 = L.seqwin[wi + 2];
                            const uint32_t sh = (uint32_t)q & 31u;
                            const uint64_t w64 = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32) | __builtin_amdgcn_alignbit(d1, d0, sh);
                            const uint32_t obits = oc ? (uint32_t)(w64 >> (64u - oc)) : 0u;
                            const uint32_t mlx = xm ? (uint32_t)((w64 << oc) >> (64u - xm)) : 0u;
                            const uint32_t llx = xl ? (uint32_t)((w64 << (oc + xm)) >> (64u - xl)) : 0u;
                            ov = (1u << oc) + obits;
                            ml = (mc < 32u ? mc + 3u : L.mltab[mc - 32u] & 0xffffffu) + mlx;
                            ll = (lc < 16u ? lc : L.lltab[lc - 16u] & 0xffffffu) + llx;
                        }
                    }
                    // ---- offsets.  A sequence with a new offset (the usual case) is done in parallel; only the ones that
                    // use the repeat-offset history are walked in order, with the history brought up to date from the
                    // up to three new offsets in front of each (values read with readlane: scalar work).
                    {
                        const uint32_t nres = cn < dec_bad ? cn : dec_bad;
                        const bool mine = lane < nres;
                        if (mine && ov > 3) off = ov - 3;
                        uint64_t repm = __ballot(mine && ov <= 3);
                        uint32_t cur = 0;  // sequences whose effect on the history is already in rep0..2
                        auto advance = [&](uint32_t to) {  // sequences cur..to-1 all carry new offsets
                            const uint32_t n = to - cur;
                            if (n >= 3) {
                                rep2 = rdlane(off, to - 3);
                                rep1 = rdlane(off, to - 2);
                                rep0 = rdlane(off, to - 1);
                            } else if (n == 2) {
                                rep2 = rep0;
                                rep1 = rdlane(off, to - 2);
                                rep0 = rdlane(off, to - 1);
                            } else if (n == 1) {
                                rep2 = rep1;
                                rep1 = rep0;
                                rep0 = rdlane(off, to - 1);
                            }
                            cur = to;
                        };
                        while (repm) {
                            const uint32_t j = (uint32_t)__ffsll((long long)repm) - 1u;
                            repm &= repm - 1;
                            advance(j);
                            const uint32_t idx = rdlane(ov, j) - (rdlane(ll, j) != 0 ? 1u : 0u);  // 3 means rep0 - 1
                            uint32_t offset = rep0;
                            if (idx != 0) {
                                uint32_t t = idx == 3 ? rep0 - 1 : (idx == 1 ? rep1 : rep2);
                                t += !t;
                                if (idx != 1) rep2 = rep1;
                                rep1 = rep0;
                                rep0 = t;
                                offset = t;
                            }
                            if (lane == j) off = offset;
                            cur = j + 1;
                        }
                        advance(nres);
                    }
                    // ---- place the chunk ---------------------------------------------------------------
                    const uint32_t lit_incl = wave_incl_scan(ll), lit_before = lit_incl - ll;
                    const uint32_t tot = ll + ml, out_incl = wave_incl_scan(tot);
                    const uint32_t ostart = opos + out_incl - tot, mstart = ostart + ll;
                    // first failing sequence decides (destination room, literal supply, capacity, offset)
                    uint32_t fail = 0;
                    if (lane == dec_bad) fail = 5;
                    else if (lane < cn && lane < dec_bad) {
                        if (dropped + ostart + tot > out_limit || (uint64_t)(ostart - block_out0) + tot > BLOCK_MAX) fail = 1;
                        else if (lit_incl > regen - lpos) fail = 2;
                        else if ((uint64_t)ostart + tot > cap) fail = 3;
                        else if (off > mstart || off > window) fail = 4;  // beyond the frame's start, or beyond its window: the verdict must not
                                                                          // depend on how much history a streaming caller has let go (api.hip, dec_compact)
                    }
                    const uint64_t failm = __ballot(fail != 0);
                    if (failm) {
                        const uint32_t f = rdlane(fail, (uint32_t)__ffsll((long long)failm) - 1);
                        if (f == 1) ZFAIL(70);
                        if (f == 3) {
                            opos = block_out0;  // whole blocks only: see include/compu_hip.h
                            status = CHIP_NEED_OUTPUT;
                            goto done;
                        }
                        ZFAIL(ZSTD_E_CORRUPTION);
                    }
                    const uint32_t LB = rdlane(lit_incl, 63), OB = rdlane(out_incl, 63);
                    // (LSYNC in both phases: the steps talk through LDS, and a wave's global stores are seen by its later loads in
                    // program order -- lanes of one wave share the CU's L1 --, so no step waits for its stores to be acknowledged)
                    // ---- phase A: all literal bytes of the chunk (their sources never depend on this chunk) ----
                    ZT_BEGIN(zt5);
#ifndef CHIP_EXP_NOEXEC
                    if (LB) {
                        LSYNC();
                        L.xpar[2 * lane] = ostart;
                        L.xpar[2 * lane + 1] = lit_before;
                        uint32_t carry = 0;
                        for (uint32_t wb = 0; wb < LB; wb += 256) {
                            ZC(9, 1);
                            L.xheads[lane] = 0;
                            LSYNC();
                            if (ll && lit_before >= wb && lit_before < wb + 256) ((uint8_t *)L.xheads)[lit_before - wb] = (uint8_t)(lane + 1);
                            LSYNC();
                            const uint32_t h = L.xheads[lane];
                            uint32_t r0 = h & 0xffu, r1 = (h >> 8) & 0xffu, r2 = (h >> 16) & 0xffu, r3 = h >> 24;
                            r1 = r1 > r0 ? r1 : r0;
                            r2 = r2 > r1 ? r2 : r1;
                            r3 = r3 > r2 ? r3 : r2;
                            uint32_t cin = wave_shr1(wave_incl_max_scan(r3));
                            cin = cin > carry ? cin : carry;
                            r0 = r0 > cin ? r0 : cin;
                            r1 = r1 > cin ? r1 : cin;
                            r2 = r2 > cin ? r2 : cin;
                            r3 = r3 > cin ? r3 : cin;
                            carry = rdlane(r3, 63);
                            const uint32_t rr[4] = {r0, r1, r2, r3};
                            uint32_t dsts[4];
                            uint8_t bytes[4];
                            bool has[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t q = wb + 4u * lane + j;
                                has[j] = q < LB && rr[j] != 0;
                                const uint32_t o = has[j] ? rr[j] - 1u : 0u;
                                dsts[j] = L.xpar[2 * o] + (q - L.xpar[2 * o + 1]);
                                bytes[j] = lit_mode == 1 ? (uint8_t)lit_rle : litsrc[has[j] ? lpos + q : 0u];
                            }
                            // the four bytes are put together before the first store: one wait for the loads, and none between
                            // the stores (a wait in front of every conditional store would also wait for the store before it)
                            uint32_t packed = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
                            asm volatile("" : "+v"(packed));  // (keeps the compiler from storing the loaded bytes one by one after all)
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (has[j]) gout[dsts[j]] = (uint8_t)(packed >> (8 * j));
                            LSYNC();
                        }
                    }
                    ZT_END(5, zt5);
                    // ---- phase B: matches, several per step as long as none reads what the step writes ----------
                    ZT_BEGIN(zt6);
                    {
                        const uint32_t mbi = wave_incl_scan(ml), mbx = mbi - ml;
                        const uint32_t srcend = mstart - off + (ml < off ? ml : off);
                        uint64_t mm = __ballot(ml != 0);
                        // A step's plan -- which matches go together, the owner of each of its bytes scattered into LDS -- depends on
                        // positions only, never on data: the plan of the next step is laid out in LDS while this step's loads are in
                        // flight (its LDS arrays are free again once this step's addresses are in registers).
                        uint64_t inc = 0;
                        uint32_t nbytes = 0, pk0 = 0;
                        bool longm = false;
                        auto plan = [&]() {  // the step that starts at mm's first match
                            ZC(10, 1);
                            const uint32_t k0 = (uint32_t)__ffsll((long long)mm) - 1;
                            const uint32_t d0 = rdlane(mstart, k0), b0 = rdlane(mbx, k0), l0 = rdlane(ml, k0);
                            pk0 = k0;
                            longm = l0 > 256;
                            if (longm) {  // a long match is copied on its own, in its turn
                                inc = 1ull << k0;
                                return;
                            }
                            const uint64_t okm = __ballot(ml && (mbi - b0 <= 256u) && (lane == k0 || srcend <= d0));
                            const uint64_t rem = mm & ~okm;
                            inc = rem ? (mm & ((1ull << ((uint32_t)__ffsll((long long)rem) - 1)) - 1ull)) : mm;
                            const uint32_t lastl = 63u - (uint32_t)__clzll((long long)inc);
                            nbytes = rdlane(mbi, lastl) - b0;
                            L.xheads[lane] = 0;
                            LSYNC();
                            if ((inc >> lane) & 1ull) {
                                const uint32_t rank = (uint32_t)__popcll(inc & lanemask_lt());
                                const uint32_t rel = mbx - b0;
                                ((uint8_t *)L.xheads)[rel] = (uint8_t)(rank + 1);
                                L.xpar[3 * rank] = mstart;
                                L.xpar[3 * rank + 1] = off;
                                L.xpar[3 * rank + 2] = (ml - 1u) | (rel << 8);
                            }
                        };
                        if (mm) plan();
                        while (mm) {
                            if (longm) {
                                wave_match_copy(gout + rdlane(mstart, pk0), rdlane(off, pk0), rdlane(ml, pk0));
                                mm &= ~inc;
                                if (mm) plan();
                                continue;
                            }
                            LSYNC();
                            const uint32_t h = L.xheads[lane];
                            uint32_t r0 = h & 0xffu, r1 = (h >> 8) & 0xffu, r2 = (h >> 16) & 0xffu, r3 = h >> 24;
                            r1 = r1 > r0 ? r1 : r0;
                            r2 = r2 > r1 ? r2 : r1;
                            r3 = r3 > r2 ? r3 : r2;
                            const uint32_t cin = wave_shr1(wave_incl_max_scan(r3));
                            r0 = r0 > cin ? r0 : cin;
                            r1 = r1 > cin ? r1 : cin;
                            r2 = r2 > cin ? r2 : cin;
                            r3 = r3 > cin ? r3 : cin;
                            const uint32_t rr[4] = {r0, r1, r2, r3};
                            uint32_t srcs[4], dsts[4];
                            bool has[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t q = 4u * lane + j;
                                has[j] = q < nbytes && rr[j] != 0;
                                const uint32_t r = has[j] ? rr[j] - 1u : 0u;
                                const uint32_t st = L.xpar[3 * r], ds = L.xpar[3 * r + 1], pv = L.xpar[3 * r + 2];
                                const uint32_t ln = (pv & 0xffu) + 1u, o = q - (pv >> 8);
                                dsts[j] = st + o;
                                srcs[j] = st - ds + (ds >= ln ? o : o % ds);
                            }
                            uint8_t bytes[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) bytes[j] = gout[has[j] ? srcs[j] : 0u];
                            LSYNC();  // (this step's reads of the plan are done: the arrays are free)
                            mm &= ~inc;
                            if (mm) plan();  // the next step's plan goes into LDS while the loads fly
                            uint32_t packed = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
                            asm volatile("" : "+v"(packed));  // (keeps the compiler from storing the loaded bytes one by one after all)
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (has[j]) gout[dsts[j]] = (uint8_t)(packed >> (8 * j));
                        }
                        LSYNC();
                    }
#endif
                    ZT_END(6, zt6);
                    ZT_END(4, zt4);
                    opos += OB;
                    lpos += LB;
                }
                if (s.avail != (int32_t)s.used) ZFAIL(ZSTD_E_CORRUPTION);  // the bitstream must be consumed exactly
            }
            const uint32_t restl = regen - lpos;
            if (dropped + opos + restl > out_limit) ZFAIL(70);
            if ((uint64_t)(opos - block_out0) + restl > BLOCK_MAX) ZFAIL(70);
            if ((uint64_t)opos + restl > cap) {
                opos = block_out0;
                status = CHIP_NEED_OUTPUT;
                goto done;
            }
            copy_literals(restl);
            opos += restl;
            ip = bend;
        }
        if (last) {
            if (has_fcs && dropped + opos != fcs) ZFAIL(ZSTD_E_CORRUPTION);
            blocks_done = true;
        }
        save_checkpoint(blocks_done);
    }
    if (has_checksum) {
        if (END - ip < 4) ZNEED_INPUT();
        uint32_t want = rd32_at(b, ip * 8u);
#ifndef CHIP_EXP_NOXXH
        uint32_t got;
        if (rs) {  // streaming: the state carried from block to block, then the bytes behind its last stripe
            uint64_t acc;
            uint32_t rel;
            hash_upto_opos((uint64_t *)L.huf, 128, acc, rel);
            got = xxh_finish(acc, dropped + opos, gout + rel, opos - rel);
        } else {
            ZT_BEGIN(zt7);
            got = wave_xxh64_low32((uint64_t *)L.huf, gout, opos);  // the Huffman table is dead by now
            ZT_END(7, zt7);
        }
        if (got != want) ZFAIL(ZSTD_E_CHECKSUM_WRONG);
#endif
        ip += 4;
    }
    status = CHIP_FINISHED;
done:
    if ((a.flags & F_COMPU_STATUS) && status != CHIP_FINISHED) {
        // compu looks at the output first (src/decoder/zstd.rs:121-133): output.pos == output.size is NeedOutput whatever
        // ZSTD_decompressStream returned.  An error return leaves output.pos as compu set it, 0 -- libzstd decodes a block only once the one
        // in front has been flushed whole, and returns from inside its loop -- so an error stays an error (also behind blocks that fill
        // the range exactly) unless the range is empty; checked against the system's libzstd in tests/test_oracle_zstd.py.
        if (status < 0) {
            if (cap == 0) status = CHIP_NEED_OUTPUT;
        } else if (opos >= cap) {
            status = CHIP_NEED_OUTPUT;
            opos = cap;
        }
    }
    if (rs && lane == 0 && status != CHIP_NEED_INPUT && status != CHIP_NEED_OUTPUT) rs[0] = 0;  // nothing to continue
    if (lane == 0) {
        a.out_len[u] = opos;
        a.in_used[u] = status == CHIP_NEED_INPUT ? in_len : ip - B0;
        a.status[u] = status;
#ifdef CHIP_STATS
        if (a.stats) {
            zst[0] = __builtin_readcyclecounter() - zt0;
            for (int i = 0; i < 24; i++) a.stats[(size_t)u * 24 + i] = zst[i];
        }
#endif
    }
#undef ZFAIL
#undef ZNEED_INPUT
}

}  // namespace

hipError_t launch_zstd_decode(const BatchArgs &a, int window_log_max, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(zstd_kernel, dim3(a.n), dim3(64), 0, stream, a, window_log_max ? window_log_max : 27);
    return hipGetLastError();
}

}  // namespace chip
