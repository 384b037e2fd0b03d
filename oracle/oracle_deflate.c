/*
 * oracle/oracle_deflate.c -- CPU restatement of the level-1 class DEFLATE encoder behind
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
 * bytes, distance <= 32768), then the table takes the highest position per slot.  Tokens are chosen
 * greedily left to right and emitted with the fixed Huffman code (RFC 1951 sec. 3.2.6).  A segment
 * whose fixed-Huffman form is not smaller than stored blocks is emitted stored.  Level 0 = stored.
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

/* One byte-aligned deflate segment for in[0..n): fixed-Huffman block (or stored blocks when not
 * larger), closed by the final-block flag (final) or by a sync marker (!final).  Returns the size;
 * *overflow is set when cap was too small. */
static size_t encode_segment(const uint8_t *in, size_t n, uint8_t *out, size_t cap, int level, int strategy, int final, int *overflow)
{
    bitw w = {out, cap, 0, 0, 0, 0};
    size_t ssz = stored_size(n, !final);
    int use_stored = level == 0;
    if (!use_stored) {
        uint32_t *table = (uint32_t *)calloc(1u << HASH_BITS, sizeof(uint32_t));
        bw_put(&w, (uint32_t)(final ? 1 : 0) | (1u << 1), 3);
        size_t skip = 0;
        for (size_t base = 0; base < n; base += 64) {
            unsigned mlen[64], mdist[64];
            uint32_t hh[64];
            for (unsigned l = 0; l < 64; l++) {
                size_t p = base + l;
                mlen[l] = 0;
                hh[l] = 0xffffffffu;
                if (p + 4 > n) continue;
                uint32_t v = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16) | ((uint32_t)in[p + 3] << 24);
                uint32_t h = (v * 2654435761u) >> (32 - HASH_BITS);
                hh[l] = h;
                uint32_t c = table[h];
                /* strategy 3 (Z_RLE): the only candidate is the byte before; 2 (Z_HUFFMAN_ONLY): none */
                if (strategy == 3) c = p > 0 ? (uint32_t)p : 0;
                if (strategy == 2) c = 0;
                if (c && p - (c - 1) <= MAX_DIST) {
                    size_t q = c - 1, lim = n - p < MAX_MATCH ? n - p : MAX_MATCH, k = 0;
                    while (k < lim && in[q + k] == in[p + k]) k++;
                    if (k >= MIN_MATCH) { mlen[l] = (unsigned)k; mdist[l] = (unsigned)(p - q); }
                }
            }
            for (unsigned l = 0; l < 64; l++)
                if (hh[l] != 0xffffffffu && table[hh[l]] < base + l + 1) table[hh[l]] = (uint32_t)(base + l + 1);
            size_t pos = skip;
            while (pos < 64 && base + pos < n) {
                int nb;
                uint32_t bits;
                if (mlen[pos] >= MIN_MATCH) { bits = match_bits(mlen[pos], mdist[pos], &nb); pos += mlen[pos]; }
                else { bits = lit_bits(in[base + pos], &nb); pos += 1; }
                bw_put(&w, bits, nb);
            }
            skip = pos > 64 ? pos - 64 : 0;
        }
        free(table);
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
        size_t bound = stored_size(s->in_len, !final) + 16;
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
