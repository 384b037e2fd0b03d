// Wave-parallel CRC-32 (RFC 1952 sec. 8) and Adler-32 (RFC 1950 sec. 9) over bytes in HBM, shared by
// the inflate and deflate kernels.  Both take the running value of a previous segment as seed.
#pragma once
#include "chip_internal.h"

namespace chip {

// ---- checksums over bytes in HBM, wave-parallel ------------------------------------------------
// x^(2^k) mod P for the reflected CRC-32 polynomial 0xEDB88320 (bit 31 = x^0)
__device__ __constant__ static const uint32_t X2N[32] = {
    0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u,
    0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu,
    0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu,
    0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu};

// a(x) * b(x) mod P, reflected representation
__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return p;
}

// Split [0,n) into 64 right-aligned chunks of `chunk` bytes (chunk = power of two >= n/64): lane i
// gets [beg,end); leading lanes may be empty and the first non-empty one may be short.
__device__ __forceinline__ void lane_chunk(uint32_t n, uint32_t &chunk_log2, uint32_t &beg, uint32_t &end)
{
    uint32_t per = (n + 63u) >> 6;
    chunk_log2 = per <= 1 ? 0u : 32u - (uint32_t)__clz((int)(per - 1));
    uint64_t chunk = 1ull << chunk_log2;
    int64_t e = (int64_t)n - (int64_t)(63 - (int64_t)lane_id()) * (int64_t)chunk;
    int64_t b = e - (int64_t)chunk;
    end = e > 0 ? (uint32_t)e : 0u;
    beg = b > 0 ? (uint32_t)b : 0u;
}

// CRC-32 (RFC 1952 sec. 8) of p[0..n).  `tab` is 2048 words (8 KB) of LDS scratch: eight 256-entry tables
// (slicing by 8: table k advances the register over a byte that lies k bytes further on).  A lane runs over its chunk
// with aligned 16-byte loads, eight bytes per dependent step; the chunks' registers are then combined across lanes.
// Out of line (three call sites in the inflate kernel; inlined it was a fifth of the kernel's code): the table pointer carries its
// address space, so its accesses stay LDS accesses.
__device__ static __attribute__((noinline, unused)) uint32_t wave_crc32(LDS_AS uint32_t *tab, const uint8_t *p_, uint32_t n, uint32_t seed = 0)
{
    const uint32_t lane = lane_id();
    const GAS uint8_t *const p = (const GAS uint8_t *)p_;
    WSYNC();
    for (uint32_t i = lane; i < 256; i += 64) {
        uint32_t c = i;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
        tab[i] = c;
    }
    WSYNC();
    for (uint32_t k = 1; k < 8; k++) {
        for (uint32_t i = lane; i < 256; i += 64) {
            const uint32_t c = tab[(k - 1) * 256 + i];
            tab[k * 256 + i] = (c >> 8) ^ tab[c & 0xffu];
        }
        WSYNC();
    }
    uint32_t lg, beg, end;
    lane_chunk(n, lg, beg, end);
    uint32_t c = (beg == 0 && end > 0) ? ~seed : 0u;  // the lane that owns byte 0 carries the running value
    if (n == 0 && lane == 63) c = ~seed;
    uint32_t k = beg;
    // bytes up to a 16-byte boundary of the address
    while (k < end && (((uintptr_t)(p + k)) & 15u) != 0) {
        c = tab[(c ^ p[k]) & 0xffu] ^ (c >> 8);
        k++;
    }
    // 16 bytes per load, two steps of eight bytes
    for (; k + 16 <= end; k += 16) {
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ d = *(const GAS u32x4_ *)(p + k);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t lo = (h ? d.z : d.x) ^ c, hi = h ? d.w : d.y;
            c = tab[7 * 256 + (lo & 0xffu)] ^ tab[6 * 256 + ((lo >> 8) & 0xffu)] ^ tab[5 * 256 + ((lo >> 16) & 0xffu)] ^ tab[4 * 256 + (lo >> 24)] ^
                tab[3 * 256 + (hi & 0xffu)] ^ tab[2 * 256 + ((hi >> 8) & 0xffu)] ^ tab[1 * 256 + ((hi >> 16) & 0xffu)] ^ tab[hi >> 24];
        }
    }
    for (; k < end; k++) c = tab[(c ^ p[k]) & 0xffu] ^ (c >> 8);
    // tree combine: state(A||B) = state(A) * x^(8|B|) + raw(B); right blocks have 2^k * chunk bytes
#pragma unroll
    for (int k2 = 0; k2 < 6; k2++) {
        uint32_t left = (uint32_t)__shfl_up((int)c, 1 << k2, 64);
        uint32_t f = X2N[(3 + lg + k2) & 31];
        uint32_t comb = multmodp(f, left) ^ c;
        if ((lane & ((2u << k2) - 1)) == ((2u << k2) - 1)) c = comb;
    }
    WSYNC();
    return ~rdlane(c, 63);
}

// Adler-32 (RFC 1950 sec. 9) of p[0..n): per lane chunk sums (16 bytes per load), then the weighted combination
__device__ inline uint32_t wave_adler32(const uint8_t *p, uint32_t n, uint32_t seed = 1)
{
    uint32_t lg, beg, end;
    lane_chunk(n, lg, beg, end);
    uint32_t a = 0, b = 0, k = beg;
    while (k < end && (((uintptr_t)(p + k)) & 15u) != 0) {
        a += p[k];
        b += a;
        k++;
    }
    while (k + 16 <= end) {
        // at most 2048 bytes between reductions: b grows by less than 2048 * (a + 255 * 2048) < 2^32 with a, b < 65521
        const uint32_t stop = end - k > 2048 ? k + 2048 : k + ((end - k) & ~15u);
        for (; k < stop; k += 16) {
            const uint4 d = *(const uint4 *)(p + k);
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t b0 = w[j] & 0xffu, b1 = (w[j] >> 8) & 0xffu, b2 = (w[j] >> 16) & 0xffu, b3 = w[j] >> 24;
                b += 4u * a + 4u * b0 + 3u * b1 + 2u * b2 + b3;
                a += b0 + b1 + b2 + b3;
            }
        }
        a %= 65521u;
        b %= 65521u;
    }
    for (; k < end; k++) {
        a += p[k];
        b += a;
    }
    a %= 65521u;
    b %= 65521u;
    // B_total = sum_i (b_i + a_i * bytes_after_i) + n ; A_total = 1 + sum a_i
    uint64_t after = (uint64_t)(n - end);
    uint32_t term = (uint32_t)(((uint64_t)b + (uint64_t)a * (after % 65521u)) % 65521u);
    uint32_t sa = wave_incl_scan(a), sb = wave_incl_scan(term);
    // continuing from (a0, b0): A = a0 + sum ; B = b0 + n * a0 + weighted sum
    const uint32_t a0 = seed & 0xffffu, b0 = seed >> 16;
    uint32_t A = (a0 + rdlane(sa, 63)) % 65521u;
    uint32_t B = (uint32_t)(((uint64_t)rdlane(sb, 63) + (uint64_t)b0 + (uint64_t)(n % 65521u) * a0) % 65521u);
    return (B << 16) | A;
}


}  // namespace chip
