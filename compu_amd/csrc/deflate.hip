// Batched DEFLATE level-1 class encoder for gfx950 (MI355X): one wavefront encodes one unit.
//
// Replaces, per unit, what compu reaches through sys::deflate (src/encoder/mod.rs:352) for an encoder
// built by Interface::zlib_ng(ZlibOptions::new().compression(1)) (src/encoder/zlib_ng.rs:50-87):
// greedy hash matching in a 32 KiB window and fixed-Huffman emission (the shape of zlib-ng's
// deflate_quick), gzip / zlib wrappers with CRC-32 / Adler-32 trailers.  compu's own tests pin the
// encoder by round trip and cross-API determinism only (tests/encoder.rs:10-78); the exact algorithm
// is stated in oracle/oracle_deflate.c and this kernel reproduces its output byte for byte.
//
// Per 64-position chunk: every lane hashes the 4 bytes at its position, looks the hash table (LDS,
// state before the chunk) up, measures the common prefix with the candidate (4-byte compares from
// HBM/L2), the table then takes the highest position per slot (ds_max), the greedy token choice is
// a scalar walk over the 64 match lengths, and the chosen tokens are packed with a wave prefix sum
// of their bit lengths into an LDS bit buffer that is flushed to HBM.
#include "chip_internal.h"
#include "wave_checksums.h"

namespace chip {

namespace {

constexpr int HASH_BITS = 12;
constexpr uint32_t MIN_MATCH = 4, MAX_MATCH = 258, MAX_DIST = 32768;
constexpr int OUT_DW = 448;  // LDS bit buffer, dwords (a chunk adds at most 62)

struct alignas(16) ELds {
    uint32_t table[1 << HASH_BITS];
    uint32_t obuf[OUT_DW + 64];
};

// 4 input bytes at byte offset `off` of the dword-aligned view (little endian)
__device__ __forceinline__ uint32_t ld32(const uint32_t *g32, uint32_t total_dw, uint32_t off)
{
    uint32_t i = off >> 2;
    uint32_t d0 = i < total_dw ? g32[i] : 0u;
    uint32_t d1 = i + 1 < total_dw ? g32[i + 1] : 0u;
    return __builtin_amdgcn_alignbit(d1, d0, (off & 3u) * 8u);
}

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t n) { return __brev(v) >> (32 - n); }

// RFC 1951 sec. 3.2.5 / 3.2.6: one token as LSB-first bits (<= 31)
__device__ __forceinline__ uint32_t lit_code(uint32_t v, uint32_t &n)
{
    if (v < 144) {
        n = 8;
        return rev_bits(0x30 + v, 8);
    }
    n = 9;
    return rev_bits(0x190 + (v - 144), 9);
}

__device__ __forceinline__ uint32_t match_code(uint32_t len, uint32_t dist, uint32_t &n)
{
    // length code: 0..7 -> lengths 3..10; then groups of four per extra-bit count; 28 -> 258
    uint32_t lc, lbase, lext;
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
    uint32_t dc, dbase, dext;
    if (dist < 5) {
        dc = dist - 1;
        dbase = dist;
        dext = 0;
    } else {
        uint32_t y = dist - 1;                             // 4..32767
        dext = (31u - (uint32_t)__clz((int)y)) - 1u;       // 1..13
        dc = 2u * dext + 2u + ((y >> dext) & 1u);
        dbase = 1u + ((2u + ((y >> dext) & 1u)) << dext);
    }
    uint32_t sym = 257 + lc, bits, nb;
    if (sym < 280) {
        bits = rev_bits(sym - 256, 7);
        nb = 7;
    } else {
        bits = rev_bits(0xC0 + (sym - 280), 8);
        nb = 8;
    }
    bits |= (len - lbase) << nb;
    nb += lext;
    bits |= rev_bits(dc, 5) << nb;
    nb += 5;
    bits |= (dist - dbase) << nb;
    nb += dext;
    n = nb;
    return bits;
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
__device__ uint32_t flush_bits(ELds &L, uint8_t *gout, uint32_t cap, uint32_t obytes, uint32_t &nbits, bool all)
{
    WSYNC();
    const uint32_t lane = lane_id();
    uint32_t nbytes = all ? (nbits + 7u) >> 3 : (nbits >> 5) << 2;  // whole dwords unless closing
    const uint8_t *src = (const uint8_t *)L.obuf;
    for (uint32_t j = lane; j < nbytes; j += 64)
        if (obytes + j < cap) gout[obytes + j] = src[j];
    WSYNC();
    // keep the partial dword, clear the rest
    uint32_t keep_w = nbytes >> 2;
    uint32_t carry = (!all && keep_w < (uint32_t)(OUT_DW + 64)) ? L.obuf[keep_w] : 0u;
    WSYNC();
    for (uint32_t j = lane; j < (uint32_t)(OUT_DW + 64); j += 64) L.obuf[j] = j == 0 ? carry : 0u;
    WSYNC();
    nbits = all ? 0u : nbits - nbytes * 8u;
    return nbytes;
}

// append `n` bits (uniform value) to the bit buffer
__device__ __forceinline__ void put_uniform(ELds &L, uint32_t &nbits, uint32_t bits, uint32_t n)
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

__global__ __launch_bounds__(64) void deflate_kernel(EncArgs a)
{
    __shared__ ELds L;
    const uint32_t u = blockIdx.x;
    if (u >= a.b.n) return;
    const uint32_t lane = lane_id();
    const uint8_t *gin = a.b.in_base + a.b.in_off[u];
    const uint32_t n = a.b.in_len[u];
    uint8_t *gout = a.b.out_base + a.b.out_off[u];
    const uint32_t cap = a.b.out_cap[u];
    const bool hdr = a.flags & 1u, trl = a.flags & 2u, final = a.flags & 4u;
    const uint32_t strategy = (a.flags >> 8) & 7u;
    const bool no_match = strategy == CHIP_STRATEGY_HUFFMAN_ONLY, rle = strategy == CHIP_STRATEGY_RLE;
    const int fmt = a.b.format;

    const uint32_t mis = (uint32_t)((uintptr_t)gin & 3u);
    const uint32_t *g32 = (const uint32_t *)(gin - mis);
    const uint32_t total_dw = (mis + n + 3u) >> 2;

    for (uint32_t j = lane; j < (1u << HASH_BITS); j += 64) L.table[j] = 0;
    for (uint32_t j = lane; j < (uint32_t)(OUT_DW + 64); j += 64) L.obuf[j] = 0;
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
        put_uniform(L, nbits, (final ? 1u : 0u) | (1u << 1), 3);
        uint32_t skip = 0;
        // Two chunks are in flight: while chunk c is measured, chosen and emitted, chunk c+1 has already done its
        // table lookups and updates (they depend on hashes and positions only, never on what matched) and its
        // candidate bytes are on their way from memory.  Every lane issues the same eleven loads per chunk (lanes
        // without a candidate compare their own position with itself) so that waiting for chunk c's bytes can leave
        // chunk c+1's in flight.
        struct Cand {
            uint32_t v, q, qd[5], pd[5];
            bool has;
        };
        const uint32_t last_dw = total_dw ? total_dw - 1u : 0u;
        uint32_t v_next = lane < n ? ld32(g32, total_dw, mis + lane) : 0u;
        auto lookup = [&](Cand &c, uint32_t base) {
            const uint32_t p = base + lane;
            const bool valid4 = p + 4 <= n && base < n;
            c.v = v_next;
            {
                const uint32_t off = mis + p + 64u, i = off >> 2;  // the chunk after this one, clamped into the unit
                const uint32_t d0 = g32[i < last_dw ? i : last_dw], d1 = g32[i + 1 < last_dw ? i + 1 : last_dw];
                v_next = p + 64 < n ? __builtin_amdgcn_alignbit(i + 1 < total_dw ? d1 : 0u, i < total_dw ? d0 : 0u, (off & 3u) * 8u) : 0u;
            }
            uint32_t h = 0;
            c.has = false;
            c.q = p;
            if (valid4) {
                h = (c.v * 2654435761u) >> (32 - HASH_BITS);
                const uint32_t t = L.table[h];
                if (t && p - (t - 1) <= MAX_DIST) {
                    c.has = true;
                    c.q = t - 1;
                }
                // Z_RLE: the only candidate is the byte before (distance 1); Z_HUFFMAN_ONLY: none
                if (rle) {
                    c.has = p > 0;
                    c.q = p > 0 ? p - 1 : p;
                }
                if (no_match) {
                    c.has = false;
                    c.q = p;
                }
            }
            const uint32_t qi = (mis + c.q) >> 2, pi = (mis + p) >> 2;
#pragma unroll
            for (uint32_t j = 0; j < 5; j++) {
                c.qd[j] = g32[qi + j < last_dw ? qi + j : last_dw];
                c.pd[j] = g32[pi + j < last_dw ? pi + j : last_dw];
            }
            LSYNC();  // every lookup saw the table as it stood before this chunk (LDS only: the loads stay in flight)
            if (valid4) atomicMax(&L.table[h], p + 1);
        };
        Cand cur;
        lookup(cur, 0);
        for (uint32_t base = 0; base < n; base += 64) {
            Cand nxt;
            lookup(nxt, base + 64);  // past the end this is an empty chunk: same loads, nothing looked up
            const uint32_t p = base + lane;
            uint32_t mlen = 0, mdist = 0;
            const uint32_t v = cur.v;
            if (cur.has) {
                const uint32_t q = cur.q;
                const uint32_t lim = n - p < MAX_MATCH ? n - p : MAX_MATCH;
                // the first 16 bytes of both sides arrived together: most candidates are decided right here
                uint32_t k = 16;
                const uint32_t qs = ((mis + q) & 3u) * 8u, ps = ((mis + p) & 3u) * 8u;
                bool diff = false;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t x = __builtin_amdgcn_alignbit(cur.qd[j + 1], cur.qd[j], qs) ^ __builtin_amdgcn_alignbit(cur.pd[j + 1], cur.pd[j], ps);
                    if (!diff && x) {
                        k = 4u * j + (((uint32_t)__ffs((int)x) - 1u) >> 3);
                        diff = true;
                    }
                }
                if (!diff) {
                    while (k < lim) {
                        uint32_t x = ld32(g32, total_dw, mis + q + k) ^ ld32(g32, total_dw, mis + p + k);
                        if (x) {
                            k += ((uint32_t)__ffs((int)x) - 1u) >> 3;
                            break;
                        }
                        k += 4;
                    }
                }
                if (k > lim) k = lim;
                if (k >= MIN_MATCH) {
                    mlen = k;
                    mdist = p - q;
                }
            }
            // greedy choice, left to right over the chunk: jump from selected match to selected match (a scalar
            // step per chosen match, not per position); everything in between is a literal
            const uint32_t lim64 = n - base < 64 ? n - base : 64;
            const uint64_t limmask = lim64 >= 64 ? ~0ull : ((1ull << lim64) - 1ull);
            const uint64_t cand = __ballot(mlen >= MIN_MATCH) & limmask;
            uint64_t sel = 0;
            uint32_t pos = skip;
            while (pos < lim64) {
                const uint64_t rest = cand & ~((1ull << pos) - 1ull);
                if (!rest) {
                    sel |= limmask & ~((1ull << pos) - 1ull);
                    pos = lim64;
                    break;
                }
                const uint32_t c = (uint32_t)__ffsll((long long)rest) - 1;
                sel |= ((c >= 63 ? ~0ull : ((2ull << c) - 1ull))) & ~((1ull << pos) - 1ull);
                pos = c + rdlane(mlen, c);
            }
            skip = pos > 64 ? pos - 64 : 0;
            const bool mine = (sel >> lane) & 1ull;
            uint32_t nb = 0, bits = 0;
            if (mine) bits = mlen >= MIN_MATCH ? match_code(mlen, mdist, nb) : lit_code(v & 0xffu, nb);
            const uint32_t incl = wave_incl_scan(nb);
            if (mine) {
                uint32_t at = nbits + incl - nb, w = at >> 5, sh = at & 31u;
                atomicOr(&L.obuf[w], bits << sh);
                if (sh && (nb + sh > 32)) atomicOr(&L.obuf[w + 1], bits >> (32 - sh));
            }
            nbits += rdlane(incl, 63);
            if (nbits > (uint32_t)(OUT_DW - 64) * 32u) obytes += flush_bits(L, gout, cap, obytes, nbits, false);
            cur = nxt;
        }
        put_uniform(L, nbits, 0, 7);  // end of block
        if (!final) {
            put_uniform(L, nbits, 0, 3);  // empty stored block = sync marker (Z_SYNC_FLUSH)
            nbits = (nbits + 7u) & ~7u;
            put_uniform(L, nbits, 0xffff0000u, 32);
        } else nbits = (nbits + 7u) & ~7u;
        obytes += flush_bits(L, gout, cap, obytes, nbits, true);
        body_bytes = obytes - hdr_bytes;
        if (ssz < body_bytes) use_stored = true;
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
    if (fmt == CHIP_FMT_GZIP) check = wave_crc32(L.table, gin, n, a.check_seed);  // the hash table is dead by now
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
        if (a.b.in_used) a.b.in_used[u] = n;
        if (a.check_out) a.check_out[u] = check;
    }
}

}  // namespace

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
    hipLaunchKernelGGL(deflate_kernel, dim3(b.n), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace chip
