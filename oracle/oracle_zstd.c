/*
 * oracle/oracle_zstd.c -- CPU restatement of the zstd frame decode compu reaches through
 * ZSTD_decompressStream (src/decoder/zstd.rs:110).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The arithmetic is RFC 8878 (Zstandard frame format).  The implementation compu binds
 * (zstd >= 1.5.5 via zstd-sys ^2.0.8, un-vendored, not in /root/reference) is restated from the
 * published format; status mapping follows src/decoder/zstd.rs:113-135.  Pinned by the reference
 * fixtures tests/golden/{10x10y,alice29.txt}.compressed.zstd and cross-checked against the
 * system libzstd in tests/test_oracle_zstd.py.
 *
 * Streaming shape: like libzstd's DStream this decoder owns buffers -- input is accumulated, whole
 * blocks are decoded into an internal history buffer (which doubles as the match window) and
 * handed to the caller as output space allows.
 */
#include "oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ZSTD_ErrorCode values (zstd_errors.h); compu reports -(code), src/decoder/zstd.rs:131 */
enum { ZE_GENERIC = 1, ZE_PREFIX_UNKNOWN = 10, ZE_FRAME_PARAM_UNSUPPORTED = 14, ZE_WINDOW_TOO_LARGE = 16,
       ZE_CORRUPTION = 20, ZE_CHECKSUM_WRONG = 22, ZE_DICT_CORRUPTED = 30, ZE_DICT_WRONG = 32, ZE_MEMORY = 64,
       ZE_DST_TOO_SMALL = 70 };

#define BLOCK_MAX (128u * 1024u)
int orc_zstd_dbg_line = 0; /* diagnostic: source line of the last corruption verdict */
#define CORRUPT() (orc_zstd_dbg_line = __LINE__, ZE_CORRUPTION)

/* ---- XXH64 (RFC 8878 sec. 3.1.1: content checksum = low 32 bits of XXH64, seed 0) ---- */
#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
#define XP3 1609587929392839161ULL
#define XP4 9650029242287828579ULL
#define XP5 2870177450012600261ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) { return rotl64(acc + in * XP2, 31) * XP1; }
static inline uint64_t xmerge(uint64_t acc, uint64_t v) { return (acc ^ xround(0, v)) * XP1 + XP4; }

uint64_t orc_xxh64(const uint8_t *p, size_t n, uint64_t seed)
{
    const uint8_t *end = p + n;
    uint64_t h;
    if (n >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        do {
            v1 = xround(v1, rd64(p));
            v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16));
            v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)n;
    while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl64(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * XP1; h = rotl64(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (*p++) * XP5; h = rotl64(h, 11) * XP1; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

/* ---- bit readers ---- */
static inline int highbit(uint32_t v) { return 31 - __builtin_clz(v); }

/* backward bitstream (RFC 8878 sec. 4.1): bits are consumed from the end of the buffer; `pos` is the
 * number of unread bits counted from the start; reads past the start yield zeros and make pos negative */
typedef struct { const uint8_t *p; int64_t pos; } bbits;

static int bb_init(bbits *b, const uint8_t *p, size_t n)
{
    if (n == 0 || p[n - 1] == 0) return -1;
    b->p = p;
    b->pos = (int64_t)(n - 1) * 8 + highbit(p[n - 1]);
    return 0;
}

static inline uint64_t bb_peek(const bbits *b, int n) /* n <= 32; MSB of the result = next bit */
{
    if (n == 0 || b->pos <= 0) return 0;
    const uint64_t mask = (1ULL << n) - 1;
    int64_t lo = b->pos - n; /* lowest bit index wanted (may be negative: zero fill) */
    size_t first = lo >= 0 ? (size_t)(lo >> 3) : 0, last = (size_t)((b->pos - 1) >> 3);
    uint64_t v = 0;
    for (size_t k = first; k <= last; k++) v |= (uint64_t)b->p[k] << (8 * (k - first));
    if (lo >= 0) return (v >> (lo & 7)) & mask;
    v &= (1ULL << b->pos) - 1; /* pos < n <= 32 here */
    return (v << (-lo)) & mask;
}
static inline uint64_t bb_read(bbits *b, int n) { uint64_t v = bb_peek(b, n); b->pos -= n; return v; }

/* forward (little-endian) bit reader for FSE table descriptions */
typedef struct { const uint8_t *p; size_t n; size_t bit; } fbits;
static inline uint32_t fb_peek(const fbits *f, int n)
{
    uint64_t v = 0;
    size_t byte = f->bit >> 3;
    for (int k = 0; k < 5; k++) v |= (uint64_t)(byte + k < f->n ? f->p[byte + k] : 0) << (8 * k);
    return (uint32_t)((v >> (f->bit & 7)) & ((1ULL << n) - 1));
}

/* ---- FSE (RFC 8878 sec. 4.1) ---- */
typedef struct { uint8_t sym; uint8_t nb; uint16_t base; } fse_e;
typedef struct { fse_e e[512]; int al; int valid; } fse_t;

/* table description -> normalized counts; returns bytes consumed or -1 */
static int fse_read_ncount(const uint8_t *p, size_t n, int max_al, int max_sym, int16_t *norm, int *al_out, int *nsym_out)
{
    if (n < 1) return -1;
    fbits f = {p, n, 0};
    int al = (int)fb_peek(&f, 4) + 5;
    f.bit += 4;
    if (al > max_al) return -1;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0, prev0 = 0;
    while (remaining > 1 && sym <= max_sym) {
        if (prev0) {
            int n0 = sym;
            while (fb_peek(&f, 2) == 3) { n0 += 3; f.bit += 2; if ((f.bit >> 3) > n) return -1; }
            n0 += (int)fb_peek(&f, 2);
            f.bit += 2;
            if (n0 > max_sym + 1) return -1;
            while (sym < n0) norm[sym++] = 0;
            if (sym > max_sym) break;
        }
        int max = (2 * threshold - 1) - remaining, count;
        uint32_t bits = fb_peek(&f, nbits);
        if ((int)(bits & (uint32_t)(threshold - 1)) < max) {
            count = (int)(bits & (uint32_t)(threshold - 1));
            f.bit += (size_t)(nbits - 1);
        } else {
            count = (int)(bits & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            f.bit += (size_t)nbits;
        }
        count--; /* -1 means "less than 1" */
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = !count;
        while (remaining < threshold) { nbits--; threshold >>= 1; }
        if ((f.bit >> 3) > n) return -1;
    }
    if (remaining != 1 || sym > max_sym + 1) return -1;
    *al_out = al;
    *nsym_out = sym;
    size_t used = (f.bit + 7) >> 3;
    if (used > n) return -1;
    return (int)used;
}

static int fse_build(fse_t *t, const int16_t *norm, int nsym, int al)
{
    int size = 1 << al, high = size - 1;
    uint16_t next[64];
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) { t->e[high--].sym = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) {
            t->e[pos].sym = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    if (pos != 0) return -1;
    for (int u = 0; u < size; u++) {
        int s = t->e[u].sym;
        uint32_t ns = next[s]++;
        int nb = al - highbit(ns);
        t->e[u].nb = (uint8_t)nb;
        t->e[u].base = (uint16_t)((ns << nb) - (uint32_t)size);
    }
    t->al = al;
    t->valid = 1;
    return 0;
}

static void fse_rle(fse_t *t, int sym) { t->e[0].sym = (uint8_t)sym; t->e[0].nb = 0; t->e[0].base = 0; t->al = 0; t->valid = 1; }

/* ---- Huffman for literals (RFC 8878 sec. 4.2) ---- */
typedef struct { uint8_t sym[2048]; uint8_t nb[2048]; int maxbits; int valid; } huf_t;

static int huf_build(huf_t *h, uint8_t *w, int n)
{
    uint32_t total = 0;
    for (int i = 0; i < n; i++) {
        if (w[i] > 11) return -1;
        if (w[i]) total += 1u << (w[i] - 1);
    }
    if (total == 0) return -1;
    int maxbits = highbit(total) + 1;
    if (maxbits > 11) return -1;
    uint32_t rest = (1u << maxbits) - total;
    if (rest & (rest - 1)) return -1; /* the implied last weight must be a power of two */
    w[n++] = (uint8_t)(highbit(rest) + 1);
    int cnt1 = 0;
    for (int i = 0; i < n; i++) cnt1 += w[i] == 1;
    if (cnt1 < 2 || (cnt1 & 1)) return -1;
    uint32_t pos = 0;
    for (int wt = 1; wt <= maxbits; wt++)
        for (int s = 0; s < n; s++)
            if (w[s] == wt) {
                uint32_t len = 1u << (wt - 1);
                for (uint32_t k = 0; k < len; k++) { h->sym[pos + k] = (uint8_t)s; h->nb[pos + k] = (uint8_t)(maxbits + 1 - wt); }
                pos += len;
            }
    h->maxbits = maxbits;
    h->valid = 1;
    return 0;
}

/* Huffman tree description; returns bytes consumed or -1 */
static int huf_read(huf_t *h, const uint8_t *p, size_t n)
{
    if (n < 1) return -1;
    uint8_t w[256];
    int nw = 0;
    unsigned hb = p[0];
    size_t used;
    if (hb >= 128) { /* direct 4-bit weights */
        nw = (int)hb - 127;
        used = 1 + (size_t)(nw + 1) / 2;
        if (used > n) return -1;
        for (int i = 0; i < nw; i++) w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
    } else { /* FSE-compressed weights, two interleaved states */
        used = 1 + hb;
        if (hb == 0 || used > n) return -1;
        int16_t norm[16];
        int al, nsym;
        int c = fse_read_ncount(p + 1, hb, 6, 12, norm, &al, &nsym);
        if (c < 0) return -1;
        fse_t t;
        if (fse_build(&t, norm, nsym, al)) return -1;
        bbits b;
        if (bb_init(&b, p + 1 + c, hb - (size_t)c)) return -1;
        uint32_t s1 = (uint32_t)bb_read(&b, al), s2 = (uint32_t)bb_read(&b, al);
        if (b.pos < 0) return -1;
        for (;;) {
            if (nw > 253) return -1;
            w[nw++] = t.e[s1].sym;
            s1 = t.e[s1].base + (uint32_t)bb_read(&b, t.e[s1].nb);
            if (b.pos < 0) { w[nw++] = t.e[s2].sym; break; }
            if (nw > 253) return -1;
            w[nw++] = t.e[s2].sym;
            s2 = t.e[s2].base + (uint32_t)bb_read(&b, t.e[s2].nb);
            if (b.pos < 0) { w[nw++] = t.e[s1].sym; break; }
        }
    }
    if (nw > 255) return -1;
    if (huf_build(h, w, nw)) return -1;
    return (int)used;
}

static int huf_stream(const huf_t *h, const uint8_t *p, size_t n, uint8_t *out, size_t nout)
{
    bbits b;
    if (bb_init(&b, p, n)) return -1;
    for (size_t i = 0; i < nout; i++) {
        uint32_t idx = (uint32_t)bb_peek(&b, h->maxbits);
        out[i] = h->sym[idx];
        b.pos -= h->nb[idx];
        if (b.pos < 0) return -1;
    }
    return b.pos == 0 ? 0 : -1; /* the stream must be consumed exactly */
}

/* ---- sequence code tables (RFC 8878 sec. 3.1.1.3.2.1.1) ---- */
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const int16_t LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const int16_t ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};

struct orc_zstd {
    int wlog_max;
    uint8_t *in;
    size_t in_len, in_cap, in_pos;
    uint8_t *out;
    size_t out_len, out_cap, delivered;
    int stage; /* 0 frame header, 1 blocks, 2 checksum, 3 frame done, 4 inside a raw block */
    size_t raw_left; int raw_last;
    uint64_t out_limit; /* libzstd sizes its output buffer min(window + block + 64, content size): a frame
                           that regenerates more than that fails with dstSize_tooSmall */
    int has_checksum, has_fcs;
    uint64_t fcs, window;
    size_t frame_start;
    huf_t huf;
    fse_t ll, of, ml;
    uint32_t rep[3];
    uint8_t *lit; /* BLOCK_MAX literal buffer */
    int err;
};

orc_zstd *orc_zstd_new(int window_log_max)
{
    orc_zstd *s = (orc_zstd *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->wlog_max = window_log_max ? window_log_max : 27; /* ZSTD_WINDOWLOG_LIMIT_DEFAULT */
    s->lit = (uint8_t *)malloc(BLOCK_MAX + 32);
    if (!s->lit) { free(s); return NULL; }
    return s;
}

void orc_zstd_reset(orc_zstd *s)
{
    s->in_len = s->in_pos = 0;
    s->out_len = s->delivered = 0;
    s->stage = 0;
    s->err = 0;
}

void orc_zstd_free(orc_zstd *s)
{
    if (!s) return;
    free(s->in); free(s->out); free(s->lit); free(s);
}

static int grow(uint8_t **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    size_t c = *cap ? *cap : 65536;
    while (c < need) c *= 2;
    uint8_t *p = (uint8_t *)realloc(*buf, c);
    if (!p) return -1;
    *buf = p;
    *cap = c;
    return 0;
}

/* one compressed block: src[0..n) -> appended to s->out.  Returns 0 or a ZSTD error code. */
static int decode_compressed_block(orc_zstd *s, const uint8_t *src, size_t n)
{
    if (n < 1) return CORRUPT();
    /* ---- literals section (sec. 3.1.1.3.1) ---- */
    unsigned b0 = src[0], type = b0 & 3, sf = (b0 >> 2) & 3;
    size_t hl, regen, comp = 0;
    int streams = 1;
    if (type < 2) {
        if (sf == 0 || sf == 2) { hl = 1; regen = b0 >> 3; }
        else if (sf == 1) { if (n < 2) return CORRUPT(); hl = 2; regen = (b0 >> 4) | ((size_t)src[1] << 4); }
        else { if (n < 3) return CORRUPT(); hl = 3; regen = (b0 >> 4) | ((size_t)src[1] << 4) | ((size_t)src[2] << 12); }
    } else {
        if (n < 5) return CORRUPT();
        uint64_t v = (uint64_t)src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24) | ((uint64_t)src[4] << 32);
        if (sf == 0 || sf == 1) { hl = 3; regen = (v >> 4) & 0x3ff; comp = (v >> 14) & 0x3ff; streams = sf ? 4 : 1; }
        else if (sf == 2) { hl = 4; regen = (v >> 4) & 0x3fff; comp = (v >> 18) & 0x3fff; streams = 4; }
        else { hl = 5; regen = (v >> 4) & 0x3ffff; comp = (v >> 22) & 0x3ffff; streams = 4; }
    }
    if (regen > BLOCK_MAX) return CORRUPT();
    const uint8_t *p = src + hl;
    size_t left = n - hl;
    if (type == 0) {
        if (regen > left) return CORRUPT();
        memcpy(s->lit, p, regen);
        p += regen; left -= regen;
    } else if (type == 1) {
        if (left < 1) return CORRUPT();
        memset(s->lit, p[0], regen);
        p += 1; left -= 1;
    } else {
        if (comp > left) return CORRUPT();
        const uint8_t *lp = p;
        size_t lleft = comp;
        if (type == 2) {
            int c = huf_read(&s->huf, lp, lleft);
            if (c < 0) return CORRUPT();
            lp += c; lleft -= (size_t)c;
        } else if (!s->huf.valid) return ZE_DICT_CORRUPTED; /* treeless without a previous table */
        if (streams == 1) {
            if (huf_stream(&s->huf, lp, lleft, s->lit, regen)) return CORRUPT();
        } else {
            if (lleft < 10) return CORRUPT();
            size_t s1 = lp[0] | ((size_t)lp[1] << 8), s2 = lp[2] | ((size_t)lp[3] << 8), s3 = lp[4] | ((size_t)lp[5] << 8);
            if (6 + s1 + s2 + s3 > lleft) return CORRUPT();
            size_t s4 = lleft - 6 - s1 - s2 - s3, seg = (regen + 3) / 4;
            if (seg * 3 > regen) return CORRUPT();
            const uint8_t *q = lp + 6;
            if (huf_stream(&s->huf, q, s1, s->lit, seg) || huf_stream(&s->huf, q + s1, s2, s->lit + seg, seg) ||
                huf_stream(&s->huf, q + s1 + s2, s3, s->lit + 2 * seg, seg) ||
                huf_stream(&s->huf, q + s1 + s2 + s3, s4, s->lit + 3 * seg, regen - 3 * seg))
                return CORRUPT();
        }
        p += comp; left -= comp;
    }
    /* ---- sequences section (sec. 3.1.1.3.2) ---- */
    if (left < 1) return CORRUPT();
    size_t nseq = p[0];
    if (nseq == 0) { p += 1; left -= 1; }
    else if (nseq < 128) { p += 1; left -= 1; }
    else if (nseq < 255) { if (left < 2) return CORRUPT(); nseq = ((nseq - 128) << 8) + p[1]; p += 2; left -= 2; }
    else { if (left < 3) return CORRUPT(); nseq = p[1] + ((size_t)p[2] << 8) + 0x7F00; p += 3; left -= 3; }
    size_t block_out0 = s->out_len, lpos = 0;
    if (grow(&s->out, &s->out_cap, s->out_len + BLOCK_MAX + 64)) return ZE_MEMORY;
    if (nseq == 0) {
        if (left != 0) return CORRUPT();
    } else {
        if (left < 1) return CORRUPT();
        unsigned modes = p[0];
        p += 1; left -= 1;
        if (modes & 3) return CORRUPT();
        fse_t *tabs[3] = {&s->ll, &s->of, &s->ml};
        const int16_t *defs[3] = {LL_DEF, OF_DEF, ML_DEF};
        const int defn[3] = {36, 29, 53}, defal[3] = {6, 5, 6}, maxal[3] = {9, 8, 9}, maxsym[3] = {35, 31, 52};
        for (int k = 0; k < 3; k++) {
            unsigned mode = (modes >> (6 - 2 * k)) & 3;
            if (mode == 0) { if (fse_build(tabs[k], defs[k], defn[k], defal[k])) return CORRUPT(); }
            else if (mode == 1) {
                if (left < 1 || p[0] > maxsym[k]) return CORRUPT();
                fse_rle(tabs[k], p[0]);
                p += 1; left -= 1;
            } else if (mode == 2) {
                int16_t norm[64];
                int al, nsym;
                int c = fse_read_ncount(p, left, maxal[k], maxsym[k], norm, &al, &nsym);
                if (c < 0 || fse_build(tabs[k], norm, nsym, al)) return CORRUPT();
                p += c; left -= (size_t)c;
            } else if (!tabs[k]->valid) return CORRUPT();
        }
        bbits b;
        if (bb_init(&b, p, left)) return CORRUPT();
        uint32_t sl = (uint32_t)bb_read(&b, s->ll.al), so = (uint32_t)bb_read(&b, s->of.al), sm = (uint32_t)bb_read(&b, s->ml.al);
        if (b.pos < 0) return CORRUPT();
        for (size_t i = 0; i < nseq; i++) {
            unsigned oc = s->of.e[so].sym, mc = s->ml.e[sm].sym, lc = s->ll.e[sl].sym;
            if (oc > 31) return CORRUPT();
            uint64_t ov = (1ULL << oc) + bb_read(&b, (int)oc);
            uint32_t mlen = ML_BASE[mc] + (uint32_t)bb_read(&b, ML_BITS[mc]);
            uint32_t llen = LL_BASE[lc] + (uint32_t)bb_read(&b, LL_BITS[lc]);
            uint64_t offset;
            if (ov > 3) {
                offset = ov - 3;
                s->rep[2] = s->rep[1]; s->rep[1] = s->rep[0]; s->rep[0] = (uint32_t)offset;
            } else {
                unsigned idx = (unsigned)ov - (llen != 0); /* 0..3; 3 means rep[0]-1 */
                if (idx == 0) offset = s->rep[0];
                else {
                    uint32_t t = idx == 3 ? s->rep[0] - 1 : s->rep[idx];
                    t += !t;
                    if (idx != 1) s->rep[2] = s->rep[1];
                    s->rep[1] = s->rep[0];
                    s->rep[0] = t;
                    offset = t;
                }
            }
            if (i + 1 < nseq) {
                sl = s->ll.e[sl].base + (uint32_t)bb_read(&b, s->ll.e[sl].nb);
                sm = s->ml.e[sm].base + (uint32_t)bb_read(&b, s->ml.e[sm].nb);
                so = s->of.e[so].base + (uint32_t)bb_read(&b, s->of.e[so].nb);
            }
            if (b.pos < 0) return CORRUPT();
            /* execute */
            /* libzstd's order of verdicts: destination room, literal supply, then the offset */
            if (s->out_len - s->frame_start + llen + mlen > s->out_limit) return ZE_DST_TOO_SMALL;
            if (s->out_len - block_out0 + llen + mlen > BLOCK_MAX) return ZE_DST_TOO_SMALL;
            if (llen > regen - lpos) return CORRUPT();
            memcpy(s->out + s->out_len, s->lit + lpos, llen);
            s->out_len += llen; lpos += llen;
            /* beyond the frame's start, or beyond its window (RFC 8878 3.1.1.1.2; libzstd takes whatever its ring still holds) */
            if (offset > s->out_len - s->frame_start || offset > s->window) return CORRUPT();
            uint8_t *d = s->out + s->out_len;
            const uint8_t *m = d - offset;
            for (uint32_t k = 0; k < mlen; k++) d[k] = m[k];
            s->out_len += mlen;
        }
        if (b.pos != 0) return CORRUPT(); /* the sequence bitstream must be consumed exactly */
    }
    size_t rest = regen - lpos;
    if (s->out_len - s->frame_start + rest > s->out_limit) return ZE_DST_TOO_SMALL;
    if (s->out_len - block_out0 + rest > BLOCK_MAX) return ZE_DST_TOO_SMALL;
    memcpy(s->out + s->out_len, s->lit + lpos, rest);
    s->out_len += rest;
    return 0;
}

/* parse as far as the accumulated input allows; returns 0 or an error code */
static int advance(orc_zstd *s)
{
    for (;;) {
        const uint8_t *p = s->in + s->in_pos;
        size_t left = s->in_len - s->in_pos;
        if (s->stage == 3) return 0;
        if (s->stage == 0) {
            if (left < 4) return 0;
            uint32_t magic = rd32(p);
            if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { /* skippable frame, sec. 3.1.2 */
                if (left < 8) return 0;
                uint64_t sz = rd32(p + 4);
                if (left < 8 + sz) return 0;
                s->in_pos += 8 + (size_t)sz;
                s->stage = 3;
                return 0;
            }
            if (magic != 0xFD2FB528u) return ZE_PREFIX_UNKNOWN;
            if (left < 5) return 0;
            unsigned fhd = p[4], fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
            static const int did_sz[4] = {0, 1, 2, 4}, fcs_sz[4] = {0, 2, 4, 8};
            size_t hsz = 5 + (single ? 0 : 1) + (size_t)did_sz[did_flag] + (size_t)(fcs_flag ? fcs_sz[fcs_flag] : (single ? 1 : 0));
            if (left < hsz) return 0;
            if (fhd & 0x08) return ZE_FRAME_PARAM_UNSUPPORTED; /* reserved bit */
            size_t q = 5;
            uint64_t window = 0;
            if (!single) {
                unsigned wd = p[q++];
                unsigned wlog = (wd >> 3) + 10;
                if (wlog > 31) return ZE_WINDOW_TOO_LARGE;
                window = 1ULL << wlog;
                window += (window >> 3) * (wd & 7);
            }
            uint32_t did = 0;
            for (int k = 0; k < did_sz[did_flag]; k++) did |= (uint32_t)p[q++] << (8 * k);
            uint64_t fcs = 0;
            int fsz = fcs_flag ? fcs_sz[fcs_flag] : (single ? 1 : 0);
            for (int k = 0; k < fsz; k++) fcs |= (uint64_t)p[q++] << (8 * k);
            if (fcs_flag == 1) fcs += 256;
            s->has_fcs = fsz > 0;
            s->fcs = fcs;
            if (single) window = fcs;
            if (window > (1ULL << s->wlog_max)) return ZE_WINDOW_TOO_LARGE;
            if (did != 0) return ZE_DICT_WRONG; /* compu never loads a dictionary */
            s->window = window;
            {
                uint64_t bm = window < BLOCK_MAX ? window : BLOCK_MAX, ring = window + bm + 64;
                s->out_limit = (s->has_fcs && fcs < ring) ? fcs : ~0ULL;
            }
            s->has_checksum = (fhd >> 2) & 1;
            s->frame_start = s->out_len;
            s->huf.valid = 0;
            s->ll.valid = s->of.valid = s->ml.valid = 0;
            s->rep[0] = 1; s->rep[1] = 4; s->rep[2] = 8;
            s->in_pos += hsz;
            s->stage = 1;
            continue;
        }
        if (s->stage == 1) {
            if (left < 3) return 0;
            uint32_t bh = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
            unsigned last = bh & 1, type = (bh >> 1) & 3;
            size_t bsz = bh >> 3;
            size_t bmax = s->window < BLOCK_MAX ? (size_t)s->window : BLOCK_MAX;
            if (type == 3) return CORRUPT();
            if (bsz > bmax) return CORRUPT(); /* Block_Maximum_Size = min(Window_Size, 128 KiB) */
            if (type == 0) { /* raw blocks stream through as their bytes arrive */
                if (s->out_len - s->frame_start + bsz > s->out_limit) return ZE_DST_TOO_SMALL;
                s->in_pos += 3;
                s->raw_left = bsz;
                s->raw_last = (int)last;
                s->stage = 4;
                continue;
            }
            size_t need = type == 1 ? 1 : bsz;
            if (left < 3 + need) return 0;
            if (type == 1) {
                if (s->out_len - s->frame_start + bsz > s->out_limit) return ZE_DST_TOO_SMALL;
                if (grow(&s->out, &s->out_cap, s->out_len + bsz)) return ZE_MEMORY;
                memset(s->out + s->out_len, p[3], bsz);
                s->out_len += bsz;
            } else {
                size_t before = s->out_len;
                int e = decode_compressed_block(s, p + 3, bsz);
                if (e) { s->out_len = before; return e; } /* libzstd hands on nothing of a block that fails */
            }
            s->in_pos += 3 + need;
            if (last) {
                if (s->has_fcs && s->out_len - s->frame_start != s->fcs) return CORRUPT();
                s->stage = s->has_checksum ? 2 : 3;
            }
            continue;
        }
        if (s->stage == 4) {
            size_t k = s->raw_left < left ? s->raw_left : left;
            if (grow(&s->out, &s->out_cap, s->out_len + k)) return ZE_MEMORY;
            memcpy(s->out + s->out_len, p, k);
            s->out_len += k;
            s->in_pos += k;
            s->raw_left -= k;
            if (s->raw_left) return 0;
            if (s->raw_last) {
                if (s->has_fcs && s->out_len - s->frame_start != s->fcs) return CORRUPT();
                s->stage = s->has_checksum ? 2 : 3;
            } else s->stage = 1;
            continue;
        }
        if (s->stage == 2) {
            if (left < 4) return 0;
            uint32_t want = rd32(p);
            uint32_t got = (uint32_t)orc_xxh64(s->out + s->frame_start, s->out_len - s->frame_start, 0);
            if (want != got) return ZE_CHECKSUM_WRONG;
            s->in_pos += 4;
            s->stage = 3;
            continue;
        }
    }
}

orc_decode_t orc_zstd_decode(orc_zstd *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    orc_decode_t r = {in_len, out_len, ORC_NEED_INPUT, 0};
    size_t taken = 0;
    if (!s->err && s->stage != 3 && in_len) {
        if (grow(&s->in, &s->in_cap, s->in_len + in_len)) { r.status = -1; r.err = -ZE_MEMORY; return r; }
        memcpy(s->in + s->in_len, in, in_len);
        s->in_len += in_len;
        taken = in_len;
    }
    if (!s->err && s->stage != 3) s->err = advance(s);
    size_t avail = s->out_len - s->delivered, n = avail < out_len ? avail : out_len;
    /* ZSTD_decompressStream decodes a block only when the one before has been flushed whole, and an error return leaves
     * output->pos as the caller set it (0 in compu's decode_fn, src/decoder/zstd.rs:104-112): the call that gets to the damage -- the
     * one that could hand on the last good bytes -- reports the error and no output at all; compu then sees pos 0 != size
     * (checked against the system's libzstd: tests/test_oracle_zstd.py).  Only an empty output range reads as NeedOutput. */
    int err_now = s->err && s->delivered + n == s->out_len;
    if (err_now && out_len) n = 0;
    memcpy(out, s->out + s->delivered, n);
    s->delivered += n;
    r.output_remain = out_len - n;
    size_t giveback = 0;
    if (s->stage == 3 || s->err) {
        size_t trailing = s->in_len - s->in_pos;
        giveback = trailing < taken ? trailing : taken;
        s->in_len -= giveback;
    }
    r.input_remain = (in_len - taken) + giveback;
    /* ZSTD_decompressStream's return value: 0 = frame done and flushed, error, or a positive hint */
    int done = s->stage == 3 && s->delivered == s->out_len;
    int is_err = err_now; /* the good bytes of earlier calls have been handed on; those of this call are not */
    /* src/decoder/zstd.rs:113-135: 0 -> Finished; else output full -> NeedOutput; else not an error
     * -> NeedInput; else Err */
    if (done) r.status = ORC_FINISHED;
    else if (r.output_remain == 0 && !(is_err && out_len)) r.status = ORC_NEED_OUTPUT;
    else if (!is_err) r.status = ORC_NEED_INPUT;
    else { r.status = -1; r.err = -s->err; }
    return r;
}

const char *orc_zstd_strerror(int32_t code)
{
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

/* ---- many independent frames ---- */
typedef struct {
    size_t lo, hi;
    const uint8_t *in_base; const uint64_t *in_off; const uint32_t *in_len;
    uint8_t *out_base; const uint64_t *out_off; const uint32_t *out_cap;
    uint32_t *out_len; int32_t *status; size_t bad;
} zjob;

static void *zworker(void *arg)
{
    zjob *j = (zjob *)arg;
    orc_zstd *s = orc_zstd_new(0);
    j->bad = 0;
    for (size_t i = j->lo; i < j->hi; i++) {
        orc_decode_t r = orc_zstd_decode(s, j->in_base + j->in_off[i], j->in_len[i], j->out_base + j->out_off[i], j->out_cap[i]);
        j->out_len[i] = (uint32_t)(j->out_cap[i] - r.output_remain);
        j->status[i] = r.err ? r.err : r.status;
        if (r.err || r.status != ORC_FINISHED) j->bad++;
        orc_zstd_reset(s);
    }
    orc_zstd_free(s);
    return NULL;
}

size_t orc_zstd_units(size_t n, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len, uint8_t *out_base,
                      const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, int32_t *status, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n && n > 0) n_threads = (int)n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    zjob *jobs = (zjob *)malloc(sizeof(zjob) * (size_t)n_threads);
    size_t bad = 0;
    for (int t = 0; t < n_threads; t++) {
        zjob *j = &jobs[t];
        j->lo = n * (size_t)t / (size_t)n_threads;
        j->hi = n * (size_t)(t + 1) / (size_t)n_threads;
        j->in_base = in_base; j->in_off = in_off; j->in_len = in_len;
        j->out_base = out_base; j->out_off = out_off; j->out_cap = out_cap;
        j->out_len = out_len; j->status = status;
        if (n_threads == 1) zworker(j);
        else pthread_create(&th[t], NULL, zworker, j);
    }
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        bad += jobs[t].bad;
    }
    free(th);
    free(jobs);
    return bad;
}
