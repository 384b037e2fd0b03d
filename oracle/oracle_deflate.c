/*
 * oracle/oracle_deflate.c -- CPU restatement of the greedy DEFLATE encoder behind
 * encoder::Interface::zlib_ng(opts) (src/encoder/zlib_ng.rs:50-92, src/encoder/mod.rs:334-370).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * compu's own tests pin the encoder by round trip and by cross-API determinism only
 * (tests/encoder.rs:10-78, 115-173): compressed bytes are "parity unpinned" against zlib-ng, whose
 * deflate_quick output is not determined by the format.  What is pinned here: the stream is valid
 * (system zlib and the inflate oracle decode it to the input), the status contract of
 * internal_zlib_impl_encode!, and -- because this file states the exact algorithm the HIP kernel
 * runs -- the GPU output byte for byte.
 *
 * Algorithm (level >= 1): greedy parse over 64-position chunks.  Every position of a chunk looks up
 * a 4096-entry hash table of 4-byte hashes as it stood BEFORE the chunk (so candidates are at least
 * one chunk back and the 64 lookups are independent), the match length is the common prefix (4..258
 * bytes, distance <= 32768), then the table takes the highest position per slot.  A slot holds the low
 * 16 bits of a position: the candidate is the nearest earlier position with those low bits (distance
 * (p - slot) mod 65536, none if that is 0, more than 32768 or more than p) -- a slot that was never
 * written (0) or not for 64 KiB names some other position, and like every candidate it only counts
 * if four bytes there match.  Tokens are chosen
 * greedily left to right.  Level 1 (and Z_FIXED): emitted with the fixed Huffman code (RFC 1951
 * sec. 3.2.6); a segment whose fixed-Huffman form is not smaller than stored blocks is emitted stored.
 * Levels 2..9: dynamic-Huffman blocks (write_block below); from level 6 on a slot keeps its two newest
 * positions and the older one is taken when its match is longer; from level 4 on the choice is lazy: a match
 * gives way to a longer one at the next position of the same chunk.  Level 0 = stored.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#define HASH_BITS 12
#define MIN_MATCH 4
#define MAX_MATCH 258
#define MAX_DIST 32768u

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

typedef struct { uint8_t *p; size_t cap, len; uint64_t acc; int nacc; int overflow; } bitw;

static void bw_put(bitw *w, uint32_t bits, int n)
{
    w->acc |= (uint64_t)bits << w->nacc;
    w->nacc += n;
    while (w->nacc >= 8) {
        if (w->len < w->cap) w->p[w->len] = (uint8_t)w->acc;
        else w->overflow = 1;
        w->len++;
        w->acc >>= 8;
        w->nacc -= 8;
    }
}
static void bw_align(bitw *w) { if (w->nacc) bw_put(w, 0, 8 - w->nacc); }

static uint32_t rev(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

/* one token as LSB-first bits: returns bits, *n = bit count (<= 31) */
static uint32_t lit_bits(unsigned v, int *n)
{
    if (v < 144) { *n = 8; return rev(0x30 + v, 8); }
    *n = 9;
    return rev(0x190 + (v - 144), 9);
}
static uint32_t match_bits(unsigned len, unsigned dist, int *n)
{
    int lc = 28, dc = 29;
    while (LBASE[lc] > len) lc--;
    while (DBASE[dc] > dist) dc--;
    uint32_t bits;
    int nb;
    unsigned sym = 257 + (unsigned)lc;
    if (sym < 280) { bits = rev(sym - 256, 7); nb = 7; }
    else { bits = rev(0xC0 + (sym - 280), 8); nb = 8; }
    bits |= (uint32_t)(len - LBASE[lc]) << nb;
    nb += LEXT[lc];
    bits |= rev((uint32_t)dc, 5) << nb;
    nb += 5;
    bits |= (uint32_t)(dist - DBASE[dc]) << nb;
    nb += DEXT[dc];
    *n = nb;
    return bits;
}

static size_t stored_size(size_t n, int sync)
{
    size_t blocks = n ? (n + 65534) / 65535 : 1;
    return n + 5 * blocks + (sync ? 5 : 0);
}

/* ---- the match finder: one 64-position chunk -> tokens ---------------------------------------- */
typedef struct { uint16_t table[1u << HASH_BITS]; uint16_t older[1u << HASH_BITS]; int ways; size_t skip; } matcher;

#define TOK_MATCH 0x80000000u
#define TOK_LEN(t) (((t) & 0xffu) + 3u)
#define TOK_DIST(t) ((((t) >> 9) & 0x7fffu) + 1u)

/* Greedy tokens of the chunk [base, base+64) of in[0..n), appended to tok (a literal is its byte, a match
 * TOK_MATCH | (dist-1) << 9 | (len-3)); a match may run past the chunk, m->skip carries the overrun. */
static unsigned chunk_tokens(matcher *m, const uint8_t *in, size_t n, size_t base, int strategy, int lazy, uint32_t *tok)
{
    unsigned mlen[64], mdist[64], nt = 0;
    uint32_t hh[64];
    uint16_t *table = m->table;
    for (unsigned l = 0; l < 64; l++) {
        size_t p = base + l;
        mlen[l] = 0;
        mdist[l] = 0;
        hh[l] = 0xffffffffu;
        if (p + 4 > n) continue;
        uint32_t v = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16) | ((uint32_t)in[p + 3] << 24);
        uint32_t h = (v * 2654435761u) >> (32 - HASH_BITS);
        hh[l] = h;
        uint32_t dist = ((uint32_t)p - table[h]) & 0xffffu;
        if (dist > MAX_DIST || dist > p) dist = 0;
        /* strategy 3 (Z_RLE): the only candidate is the byte before; 2 (Z_HUFFMAN_ONLY): none */
        if (strategy == 3) dist = p > 0 ? 1 : 0;
        if (strategy == 2) dist = 0;
        if (dist) {
            size_t q = p - dist, lim = n - p < MAX_MATCH ? n - p : MAX_MATCH, k = 0;
            while (k < lim && in[q + k] == in[p + k]) k++;
            if (k >= MIN_MATCH) { mlen[l] = (unsigned)k; mdist[l] = (unsigned)(p - q); }
        }
        if (m->ways == 2 && strategy != 3 && strategy != 2) { /* the slot's older position: taken only with a longer match */
            uint32_t d1 = ((uint32_t)p - table[h]) & 0xffffu, d2 = ((uint32_t)p - m->older[h]) & 0xffffu;
            if (d2 != 0 && d2 <= MAX_DIST && d2 <= p && d2 != d1) {
                size_t q = p - d2, lim = n - p < MAX_MATCH ? n - p : MAX_MATCH, k = 0;
                while (k < lim && in[q + k] == in[p + k]) k++;
                if (k >= MIN_MATCH && k > mlen[l]) { mlen[l] = (unsigned)k; mdist[l] = (unsigned)(p - q); }
            }
        }
    }
    for (unsigned l = 0; l < 64; l++)
        if (hh[l] != 0xffffffffu) { /* ascending: the highest position stands, the one before it becomes the older way */
            m->older[hh[l]] = table[hh[l]];
            table[hh[l]] = (uint16_t)(base + l);
        }
    size_t pos = m->skip;
    while (pos < 64 && base + pos < n) {
        /* lazy (levels 4..9, like zlib from level 4 on): a match gives way to a longer one at the next position of the chunk */
        const int defer = lazy && pos + 1 < 64 && base + pos + 1 < n && mlen[pos + 1] > mlen[pos];
        if (mlen[pos] >= MIN_MATCH && !defer) { tok[nt++] = TOK_MATCH | ((mdist[pos] - 1) << 9) | (mlen[pos] - 3); pos += mlen[pos]; }
        else { tok[nt++] = in[base + pos]; pos += 1; }
    }
    m->skip = pos > 64 ? pos - 64 : 0;
    return nt;
}

/* ---- dynamic Huffman blocks (levels 2..9) ------------------------------------------------------
 * Blocks close after TOK_BLOCK - 64 tokens or more (65 536: a 64 KiB unit is one block; zlib's lit_bufsize is
 * 16 384 at memLevel 8, 32 768 at 9).
 * Code lengths: plain Huffman over (frequency, symbol)-sorted leaves with the two-queue method (ties
 * take the leaf); when the deepest leaf exceeds the limit every frequency is halved (rounding up)
 * and the tree rebuilt.  Canonical codes and the code-length header follow RFC 1951 sec. 3.2.2 / 3.2.7
 * with a greedy run-length pass.  Per block the cheapest of dynamic, fixed and stored is written. */
#define TOK_BLOCK 65536u
static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static void build_lengths(const uint32_t *freq_in, int n, int maxbits, uint8_t *len)
{
    uint32_t freq[288], key[288], w[576];
    uint16_t parent[576];
    int m = 0;
    for (int i = 0; i < n; i++) { freq[i] = freq_in[i]; len[i] = 0; if (freq[i]) m++; }
    if (m == 0) return;
    if (m == 1) { for (int i = 0; i < n; i++) if (freq[i]) len[i] = 1; return; }
    for (;;) {
        /* leaves by (frequency, symbol) */
        int k = 0;
        for (int i = 0; i < n; i++) if (freq[i]) key[k++] = (freq[i] << 9) | (uint32_t)i;
        for (int i = 1; i < m; i++) { uint32_t x = key[i]; int j = i; while (j > 0 && key[j - 1] > x) { key[j] = key[j - 1]; j--; } key[j] = x; }
        for (int i = 0; i < m; i++) w[i] = key[i] >> 9;
        int li = 0, ii = m, nn = m;
        while (nn < 2 * m - 1) {
            int a = (li < m && (ii >= nn || w[li] <= w[ii])) ? li++ : ii++;
            int b = (li < m && (ii >= nn || w[li] <= w[ii])) ? li++ : ii++;
            w[nn] = w[a] + w[b];
            parent[a] = parent[b] = (uint16_t)nn;
            nn++;
        }
        /* depth replaces weight, root first */
        w[2 * m - 2] = 0;
        unsigned deepest = 0;
        for (int i = 2 * m - 3; i >= 0; i--) { w[i] = w[parent[i]] + 1; if (i < m && w[i] > deepest) deepest = w[i]; }
        if (deepest <= (unsigned)maxbits) {
            for (int i = 0; i < m; i++) len[key[i] & 511u] = (uint8_t)w[i];
            return;
        }
        for (int i = 0; i < n; i++) if (freq[i]) freq[i] = (freq[i] + 1) >> 1;
    }
}

/* canonical codes (RFC 1951 sec. 3.2.2), stored bit-reversed for the LSB-first writer */
static void canon_codes(const uint8_t *len, int n, uint16_t *code)
{
    unsigned count[16] = {0}, next[16];
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    unsigned c = 0;
    for (int b = 1; b < 16; b++) { c = (c + count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; i++) code[i] = len[i] ? (uint16_t)rev(next[len[i]]++, len[i]) : 0;
}

static int len_code(unsigned len) { int lc = 28; while (LBASE[lc] > len) lc--; return lc; }
static int dist_code(unsigned dist) { int dc = 29; while (DBASE[dc] > dist) dc--; return dc; }

static void fixed_lengths(uint8_t *ll, uint8_t *dl)
{
    for (int i = 0; i < 288; i++) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    for (int i = 0; i < 32; i++) dl[i] = i < 30 ? 5 : 0;
}

/* stored form of `len` bytes written at bit position `at` of the stream: bits */
static uint64_t stored_bits(uint64_t at, size_t len)
{
    uint64_t p = at;
    size_t off = 0;
    do {
        size_t k = len - off < 65535 ? len - off : 65535;
        p += 3;
        p = (p + 7) & ~(uint64_t)7;
        p += 32 + 8 * (uint64_t)k;
        off += k;
    } while (off < len);
    return p - at;
}

/* One block: tokens tok[0..nt) covering in[from..to), `last` = BFINAL. */
static void write_block(bitw *w, const uint32_t *tok, unsigned nt, const uint8_t *in, size_t from, size_t to, int last)
{
    uint32_t lfreq[288] = {0}, dfreq[32] = {0};
    uint64_t xbits = 0, nmatch = 0;
    for (unsigned i = 0; i < nt; i++) {
        if (tok[i] & TOK_MATCH) {
            int lc = len_code(TOK_LEN(tok[i])), dc = dist_code(TOK_DIST(tok[i]));
            lfreq[257 + lc]++;
            dfreq[dc]++;
            xbits += LEXT[lc] + DEXT[dc];
            nmatch++;
        } else lfreq[tok[i]]++;
    }
    lfreq[256]++;
    uint8_t ll[288], dl[32] = {0}, fl[288], fd[32];
    build_lengths(lfreq, 286, 15, ll);
    ll[286] = ll[287] = 0;
    build_lengths(dfreq, 30, 15, dl);
    if (!nmatch) dl[0] = 1; /* one distance code is always described */
    int hlit = 286, hdist = 30;
    while (hlit > 257 && !ll[hlit - 1]) hlit--;
    while (hdist > 1 && !dl[hdist - 1]) hdist--;
    /* code-length sequence, greedy runs */
    uint8_t all[320];
    uint16_t seq[320]; /* symbol | extra << 8 */
    int na = 0, ns = 0;
    for (int i = 0; i < hlit; i++) all[na++] = ll[i];
    for (int i = 0; i < hdist; i++) all[na++] = dl[i];
    uint32_t cfreq[19] = {0};
    uint64_t cl_extra = 0;
    for (int i = 0; i < na;) {
        int v = all[i], run = 1;
        while (i + run < na && all[i + run] == v) run++;
        i += run;
        if (v == 0) {
            while (run >= 11) { int r = run < 138 ? run : 138; seq[ns++] = (uint16_t)(18 | ((r - 11) << 8)); cl_extra += 7; run -= r; }
            if (run >= 3) { seq[ns++] = (uint16_t)(17 | ((run - 3) << 8)); cl_extra += 3; run = 0; }
            while (run-- > 0) seq[ns++] = 0;
        } else {
            seq[ns++] = (uint16_t)v;
            run--;
            while (run >= 3) { int r = run < 6 ? run : 6; seq[ns++] = (uint16_t)(16 | ((r - 3) << 8)); cl_extra += 2; run -= r; }
            while (run-- > 0) seq[ns++] = (uint16_t)v;
        }
    }
    for (int i = 0; i < ns; i++) cfreq[seq[i] & 0xff]++;
    uint8_t cl[19];
    build_lengths(cfreq, 19, 7, cl);
    int hclen = 19;
    while (hclen > 4 && !cl[CL_ORDER[hclen - 1]]) hclen--;
    /* costs */
    fixed_lengths(fl, fd);
    uint64_t dyn = 3 + 14 + 3 * (uint64_t)hclen + cl_extra + xbits, fix = 3 + xbits;
    for (int i = 0; i < 19; i++) dyn += (uint64_t)cfreq[i] * cl[i];
    for (int i = 0; i < 286; i++) { dyn += (uint64_t)lfreq[i] * ll[i]; fix += (uint64_t)lfreq[i] * fl[i]; }
    for (int i = 0; i < 30; i++) { dyn += (uint64_t)dfreq[i] * dl[i]; fix += (uint64_t)dfreq[i] * fd[i]; }
    const int use_dyn = dyn < fix;
    const uint64_t huff = use_dyn ? dyn : fix, at = (uint64_t)w->len * 8 + (uint64_t)w->nacc;
    if (stored_bits(at, to - from) < huff) {
        size_t off = from;
        do {
            size_t k = to - off < 65535 ? to - off : 65535;
            bw_put(w, (uint32_t)(last && off + k == to), 3);
            bw_align(w);
            bw_put(w, (uint32_t)k | ((uint32_t)(~k & 0xffff) << 16), 32);
            for (size_t i = 0; i < k; i++) bw_put(w, in[off + i], 8);
            off += k;
        } while (off < to);
        return;
    }
    uint16_t lc_[288], dc_[32], cc_[19];
    const uint8_t *L = use_dyn ? ll : fl, *D = use_dyn ? dl : fd;
    canon_codes(L, 288, lc_);
    canon_codes(D, 30, dc_);
    bw_put(w, (uint32_t)(last ? 1 : 0) | ((use_dyn ? 2u : 1u) << 1), 3);
    if (use_dyn) {
        canon_codes(cl, 19, cc_);
        bw_put(w, (uint32_t)(hlit - 257), 5);
        bw_put(w, (uint32_t)(hdist - 1), 5);
        bw_put(w, (uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; i++) bw_put(w, cl[CL_ORDER[i]], 3);
        for (int i = 0; i < ns; i++) {
            int sym = seq[i] & 0xff;
            bw_put(w, cc_[sym], cl[sym]);
            if (sym >= 16) bw_put(w, (uint32_t)(seq[i] >> 8), sym == 16 ? 2 : sym == 17 ? 3 : 7);
        }
    }
    for (unsigned i = 0; i < nt; i++) {
        if (tok[i] & TOK_MATCH) {
            unsigned len = TOK_LEN(tok[i]), dist = TOK_DIST(tok[i]);
            int lc = len_code(len), dc = dist_code(dist);
            bw_put(w, lc_[257 + lc], L[257 + lc]);
            bw_put(w, len - LBASE[lc], LEXT[lc]);
            bw_put(w, dc_[dc], D[dc]);
            bw_put(w, dist - DBASE[dc], DEXT[dc]);
        } else bw_put(w, lc_[tok[i]], L[tok[i]]);
    }
    bw_put(w, lc_[256], L[256]);
}

static size_t encode_segment_dynamic(const uint8_t *in, size_t n, uint8_t *out, size_t cap, int level, int strategy, int final, int *overflow)
{
    bitw w = {out, cap, 0, 0, 0, 0};
    matcher *m = (matcher *)calloc(1, sizeof *m);
    m->ways = level >= 6 ? 2 : 1;
    uint32_t *tok = (uint32_t *)malloc(TOK_BLOCK * sizeof(uint32_t));
    unsigned nt = 0;
    size_t from = 0, base = 0;
    for (;;) {
        if (base < n) { nt += chunk_tokens(m, in, n, base, strategy, level >= 4, tok + nt); base += 64; }
        const int ended = base >= n;
        if (ended || nt > TOK_BLOCK - 64) {
            size_t to = ended ? n : base + m->skip;
            write_block(&w, tok, nt, in, from, to, final && ended);
            from = to;
            nt = 0;
            if (ended) break;
        }
    }
    if (!final) {
        bw_put(&w, 0, 3); /* empty stored block = sync marker (Z_SYNC_FLUSH) */
        bw_align(&w);
        bw_put(&w, 0xffff0000u, 32);
    } else bw_align(&w);
    free(tok);
    free(m);
    *overflow = w.overflow;
    return w.len;
}

/* One byte-aligned deflate segment for in[0..n): level 1 (and Z_FIXED) = one fixed-Huffman block, or stored blocks when
 * those are smaller; levels 2..9 = dynamic-Huffman blocks; closed by the final-block flag (final) or by a sync
 * marker (!final).  Returns the size; *overflow is set when cap was too small. */
static size_t encode_segment(const uint8_t *in, size_t n, uint8_t *out, size_t cap, int level, int strategy, int final, int *overflow)
{
    if (level >= 2 && strategy != 4) return encode_segment_dynamic(in, n, out, cap, level, strategy, final, overflow);
    bitw w = {out, cap, 0, 0, 0, 0};
    size_t ssz = stored_size(n, !final);
    int use_stored = level == 0;
    if (!use_stored) {
        matcher *m = (matcher *)calloc(1, sizeof *m);
        bw_put(&w, (uint32_t)(final ? 1 : 0) | (1u << 1), 3);
        for (size_t base = 0; base < n; base += 64) {
            uint32_t tok[64];
            unsigned nt = chunk_tokens(m, in, n, base, strategy, 0, tok);
            for (unsigned i = 0; i < nt; i++) {
                int nb;
                uint32_t bits = (tok[i] & TOK_MATCH) ? match_bits(TOK_LEN(tok[i]), TOK_DIST(tok[i]), &nb) : lit_bits(tok[i], &nb);
                bw_put(&w, bits, nb);
            }
        }
        free(m);
        bw_put(&w, 0, 7); /* end of block */
        if (!final) {
            bw_put(&w, 0, 3); /* empty stored block = sync marker (Z_SYNC_FLUSH) */
            bw_align(&w);
            bw_put(&w, 0xffff0000u, 32);
        } else bw_align(&w);
        if (ssz < w.len) use_stored = 1;
    }
    if (use_stored) {
        bitw s = {out, cap, 0, 0, 0, 0};
        size_t off = 0;
        do {
            size_t k = n - off < 65535 ? n - off : 65535;
            int last = final && off + k == n;
            bw_put(&s, (uint32_t)last, 8);
            bw_put(&s, (uint32_t)k | ((uint32_t)(~k & 0xffff) << 16), 32);
            for (size_t i = 0; i < k; i++) bw_put(&s, in[off + i], 8);
            off += k;
        } while (off < n);
        if (!final) { bw_put(&s, 0, 8); bw_put(&s, 0xffff0000u, 32); }
        w = s;
    }
    *overflow = w.overflow;
    return w.len;
}

struct orc_deflate {
    int mode, level, strategy;
    uint8_t *in; size_t in_len, in_cap;    /* input not yet compressed */
    uint8_t *out; size_t out_len, out_cap, delivered;
    int started, finished;
    uint32_t check; uint64_t total_in;
};

orc_deflate *orc_deflate_new(int mode, int level)
{
    if (mode != ORC_MODE_DEFLATE && mode != ORC_MODE_ZLIB && mode != ORC_MODE_GZIP) return NULL;
    if (level < 0 || level > 9) return NULL;
    orc_deflate *s = (orc_deflate *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->mode = mode;
    s->level = level;
    orc_deflate_reset(s);
    return s;
}

/* ZlibStrategy (src/encoder/zlib_common.rs:5-24): 0 Default, 1 Filtered, 2 HuffmanOnly, 3 Rle, 4 Fixed */
int orc_deflate_set_strategy(orc_deflate *s, int strategy)
{
    if (!s || strategy < 0 || strategy > 4) return -1;
    s->strategy = strategy;
    return 0;
}

void orc_deflate_reset(orc_deflate *s)
{
    s->in_len = 0;
    s->out_len = s->delivered = 0;
    s->started = s->finished = 0;
    s->check = s->mode == ORC_MODE_ZLIB ? 1 : 0;
    s->total_in = 0;
}

void orc_deflate_free(orc_deflate *s)
{
    if (!s) return;
    free(s->in);
    free(s->out);
    free(s);
}

static int dgrow(uint8_t **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    size_t c = *cap ? *cap : 4096;
    while (c < need) c *= 2;
    uint8_t *p = (uint8_t *)realloc(*buf, c);
    if (!p) return -1;
    *buf = p;
    *cap = c;
    return 0;
}

static void out_bytes(orc_deflate *s, const uint8_t *b, size_t n)
{
    dgrow(&s->out, &s->out_cap, s->out_len + n);
    memcpy(s->out + s->out_len, b, n);
    s->out_len += n;
}

/* internal_zlib_impl_encode!, src/encoder/mod.rs:334-370 */
orc_encode_t orc_deflate_encode(orc_deflate *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int op)
{
    orc_encode_t r = {in_len, out_len, ORC_ENC_ERROR};
    if (op < ORC_OP_PROCESS || op > ORC_OP_FINISH) return r;
    size_t taken = 0;
    if (!s->finished && in_len) {
        if (dgrow(&s->in, &s->in_cap, s->in_len + in_len)) return r;
        memcpy(s->in + s->in_len, in, in_len);
        s->in_len += in_len;
        taken = in_len;
    }
    if (!s->finished && (op == ORC_OP_FLUSH || op == ORC_OP_FINISH) && (s->in_len || op == ORC_OP_FINISH || !s->started)) {
        if (!s->started) {
            if (s->mode == ORC_MODE_GZIP) {
                /* RFC 1952 sec. 2.3: no name/time, XFL 4 = fastest for level 1, OS 3 = Unix */
                uint8_t h[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, (uint8_t)(s->level == 9 ? 2 : s->level == 1 ? 4 : 0), 3};
                out_bytes(s, h, 10);
            } else if (s->mode == ORC_MODE_ZLIB) {
                /* RFC 1950 sec. 2.2: CM 8, CINFO 7; FLEVEL 0 (fastest) .. 3 by level like zlib */
                unsigned flevel = s->level < 2 ? 0 : s->level < 6 ? 1 : s->level == 6 ? 2 : 3;
                unsigned hdr = (0x78u << 8) | (flevel << 6);
                hdr += 31 - hdr % 31;
                uint8_t h[2] = {(uint8_t)(hdr >> 8), (uint8_t)hdr};
                out_bytes(s, h, 2);
            }
            s->started = 1;
        }
        int final = op == ORC_OP_FINISH, ovf = 0;
        /* dynamic levels: a block holds at least TOK_BLOCK - 64 tokens and falls back to stored (at most 6 bytes over) */
        size_t bound = stored_size(s->in_len, !final) + 6 * (s->in_len / (TOK_BLOCK - 64) + 1) + 16;
        dgrow(&s->out, &s->out_cap, s->out_len + bound);
        s->out_len += encode_segment(s->in, s->in_len, s->out + s->out_len, bound, s->level, s->strategy, final, &ovf);
        if (s->mode == ORC_MODE_GZIP) s->check = orc_crc32(s->check, s->in, s->in_len);
        else if (s->mode == ORC_MODE_ZLIB) s->check = orc_adler32(s->check, s->in, s->in_len);
        s->total_in += s->in_len;
        s->in_len = 0;
        if (final) {
            if (s->mode == ORC_MODE_GZIP) {
                uint32_t t = (uint32_t)s->total_in;
                uint8_t tr[8] = {(uint8_t)s->check, (uint8_t)(s->check >> 8), (uint8_t)(s->check >> 16), (uint8_t)(s->check >> 24),
                                 (uint8_t)t, (uint8_t)(t >> 8), (uint8_t)(t >> 16), (uint8_t)(t >> 24)};
                out_bytes(s, tr, 8);
            } else if (s->mode == ORC_MODE_ZLIB) {
                uint8_t tr[4] = {(uint8_t)(s->check >> 24), (uint8_t)(s->check >> 16), (uint8_t)(s->check >> 8), (uint8_t)s->check};
                out_bytes(s, tr, 4);
            }
            s->finished = 1;
        }
    }
    size_t avail = s->out_len - s->delivered, k = avail < out_len ? avail : out_len;
    memcpy(out, s->out + s->delivered, k);
    s->delivered += k;
    r.input_remain = in_len - taken;
    r.output_remain = out_len - k;
    if (s->delivered == s->out_len) s->out_len = s->delivered = 0;
    /* deflate() return code -> EncodeStatus, src/encoder/mod.rs:357-367 */
    if (op == ORC_OP_FINISH) r.status = (s->finished && s->out_len == 0) ? ORC_ENC_FINISHED : ORC_ENC_NEED_OUTPUT;
    else r.status = (taken || k) ? ORC_ENC_CONTINUE : ORC_ENC_NEED_OUTPUT; /* Z_OK -> Continue, Z_BUF_ERROR -> NeedOutput */
    return r;
}
