/*
 * oracle/oracle_inflate.c -- CPU restatement of the inflate path compu reaches through
 * sys::inflate (src/decoder/mod.rs:470).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The arithmetic is RFC 1951 (deflate), RFC 1950 (zlib wrapper), RFC 1952 (gzip wrapper).
 * The third-party implementation compu binds (zlib-ng 2.x via libz-ng-sys ^1.1.9) is not in
 * /root/reference; this file restates the published format and the zlib API contract
 * (/usr/include/zlib.h:502-520: Z_OK / Z_BUF_ERROR / Z_STREAM_END) and is pinned by the
 * reference fixtures in tests/golden/ and by system zlib 1.2.11 in tests/test_oracle_inflate.py.
 *
 * Resumable at token granularity: a token (literal, or length+distance pair, <= 48 bits) is
 * decoded atomically from a 64-bit bit accumulator; if the input ends inside a token the bits
 * stay in the accumulator and the call reports "all input consumed" exactly as zlib does.
 */
#include "oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define Z_OK 0
#define Z_STREAM_END 1
#define Z_NEED_DICT 2
#define Z_DATA_ERROR (-3)
#define Z_BUF_ERROR (-5)

/* table entry: [3:0] code length (0 = longer than the root, resolve canonically),
 * [7:4] extra-bit count, [9:8] kind, [31:16] base value */
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_BAD = 3 };
#define ENTRY(cl, eb, kind, base) ((uint32_t)(cl) | ((uint32_t)(eb) << 4) | ((uint32_t)(kind) << 8) | ((uint32_t)(base) << 16))
#define E_CL(e) ((e)&15u)
#define E_EB(e) (((e) >> 4) & 15u)
#define E_KIND(e) (((e) >> 8) & 3u)
#define E_BASE(e) ((e) >> 16)

enum { T_CODES = 0, T_LENS = 1, T_DISTS = 2 };

/* RFC 1951 sec. 3.2.5 */
static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
/* RFC 1951 sec. 3.2.7 */
static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

#define LIT_ROOT 10
#define DIST_ROOT 9
#define CL_ROOT 7

typedef struct {
    uint32_t lut[1 << LIT_ROOT];
    uint32_t sorted[288];
    uint32_t limit15[17]; /* limit15[l]: end (exclusive) of the 15-bit-aligned code space of lengths <= l; [0] = 0 */
    uint16_t offs[17];    /* index in sorted[] of the first symbol of length l */
    int root, maxlen;
} huff_t;

static inline unsigned bitrev15(unsigned v)
{
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0f0fu) << 4) | ((v >> 4) & 0x0f0fu);
    v = ((v & 0x00ffu) << 8) | ((v >> 8) & 0x00ffu);
    return v >> 1; /* 16-bit reverse -> keep the top 15 */
}

static uint32_t make_entry(int type, unsigned sym, unsigned len)
{
    if (type == T_CODES) return ENTRY(len, 0, K_LIT, sym);
    if (type == T_LENS) {
        if (sym < 256) return ENTRY(len, 0, K_LIT, sym);
        if (sym == 256) return ENTRY(len, 0, K_EOB, 0);
        if (sym < 286) return ENTRY(len, LEXT[sym - 257], K_LEN, LBASE[sym - 257]);
        return ENTRY(len, 0, K_BAD, 0);
    }
    if (sym < 30) return ENTRY(len, DEXT[sym], K_LEN, DBASE[sym]);
    return ENTRY(len, 0, K_BAD, 0);
}

/* canonical lookup of the code whose MSB-first 15-bit-aligned value is x15 */
static inline uint32_t huff_canon(const huff_t *h, unsigned x15)
{
    for (int l = 1; l <= 15; l++)
        if (x15 < h->limit15[l]) return h->sorted[h->offs[l] + ((x15 - h->limit15[l - 1]) >> (15 - l))];
    return ENTRY(h->maxlen ? h->maxlen : 1, 0, K_BAD, 0); /* unused code space of an incomplete set */
}

/* Canonical Huffman table from code lengths (RFC 1951 sec. 3.2.2) with zlib's acceptance rules:
 * over-subscribed -> error; incomplete -> error unless it is a lens/dists set whose longest
 * code is 1 bit; an empty set is accepted and every lookup in it is invalid. */
static int huff_build(huff_t *h, const uint8_t *lens, int n, int root, int type)
{
    unsigned count[17] = {0}, next[17];
    for (int i = 0; i < n; i++) count[lens[i]]++;
    int max = 15;
    while (max >= 1 && count[max] == 0) max--;
    h->root = root;
    h->maxlen = max;
    memset(h->limit15, 0, sizeof h->limit15);
    memset(h->offs, 0, sizeof h->offs);
    if (max == 0) {
        /* zlib hands back a 1-bit table of invalid-code markers; in the code-length-code state
         * it reads val=0/bits=1 from it without looking at the marker */
        uint32_t e = type == T_CODES ? ENTRY(1, 0, K_LIT, 0) : ENTRY(1, 0, K_BAD, 0);
        for (int i = 0; i < (1 << root); i++) h->lut[i] = e;
        h->sorted[0] = e; /* every lut entry has a non-zero length, so the canonical path is never taken */
        return 0;
    }
    int left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= (int)count[l];
        if (left < 0) return -1;
    }
    if (left > 0 && (type == T_CODES || max != 1)) return -1;
    unsigned code = 0, off = 0;
    for (int l = 1; l <= 15; l++) {
        h->offs[l] = (uint16_t)off;
        next[l] = off;
        off += count[l];
        code += count[l];
        h->limit15[l] = code << (15 - l);
        code <<= 1;
    }
    for (int s = 0; s < n; s++)
        if (lens[s]) h->sorted[next[lens[s]]++] = make_entry(type, (unsigned)s, lens[s]);
    for (unsigned idx = 0; idx < (1u << root); idx++) {
        unsigned x15 = bitrev15(idx); /* idx occupies the low `root` bits -> top of the 15 */
        uint32_t e;
        if (x15 < h->limit15[root]) e = huff_canon(h, x15);
        else if (x15 >= h->limit15[15]) e = ENTRY(max, 0, K_BAD, 0);
        else e = 0; /* longer than root */
        h->lut[idx] = e;
    }
    return 0;
}

static inline uint32_t huff_lookup(const huff_t *h, uint64_t bitsv)
{
    uint32_t e = h->lut[bitsv & ((1u << h->root) - 1)];
    if (E_CL(e) == 0) e = huff_canon(h, bitrev15((unsigned)(bitsv & 0x7fff)));
    return e;
}

/* ---- checksums ---- */
static uint32_t crc_tab[8][256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void)
{
    for (unsigned i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_tab[0][i] = c;
    }
    for (unsigned i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++) crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xff];
}

uint32_t orc_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    pthread_once(&crc_once, crc_init);
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= c;
        c = crc_tab[7][a & 0xff] ^ crc_tab[6][(a >> 8) & 0xff] ^ crc_tab[5][(a >> 16) & 0xff] ^ crc_tab[4][a >> 24] ^
            crc_tab[3][b & 0xff] ^ crc_tab[2][(b >> 8) & 0xff] ^ crc_tab[1][(b >> 16) & 0xff] ^ crc_tab[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}

uint32_t orc_adler32(uint32_t adler, const uint8_t *p, size_t n)
{
    uint32_t a = adler & 0xffff, b = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        while (k--) {
            a += *p++;
            b += a;
        }
        a %= 65521;
        b %= 65521;
    }
    return (b << 16) | a;
}

/* ---- the resumable inflater ---- */
enum {
    ST_HEAD, ST_ZDICT, ST_GZ_FIXED, ST_GZ_EXLEN, ST_GZ_EXTRA, ST_GZ_NAME, ST_GZ_COMMENT, ST_GZ_HCRC,
    ST_BLOCK, ST_STORED_HDR, ST_STORED_COPY, ST_DYN_HDR, ST_DYN_CLENS, ST_DYN_LENS, ST_CODES,
    ST_CHECK, ST_LENGTH, ST_DONE, ST_BAD, ST_DICT
};

#define WSIZE 32768u

struct orc_inflate {
    int mode;  /* windowBits as given */
    int wrap;  /* resolved wrapper: 0 raw, 1 zlib, 2 gzip */
    int st;
    uint64_t hold;
    int bits;
    int last;
    unsigned gz_flags, gz_idx, gz_xlen;
    uint32_t hcrc;  /* running crc of the gzip header (FHCRC) */
    uint32_t check; /* running crc32 / adler32 of the output */
    uint64_t total_out;
    unsigned stored_left;
    unsigned nlen, ndist, ncode, have;
    uint8_t lens[320];
    huff_t clh, lith, dsth;
    int fixed_ready;
    huff_t fixl, fixd;
    unsigned pend_len, pend_dist; /* match remainder waiting for output space */
    int pend_lit;                 /* literal waiting for output space, -1 none */
    uint8_t win[WSIZE];
    unsigned wnext, whave;
    const char *msg;
    int bad_code;
};

orc_inflate *orc_inflate_new(int mode)
{
    if (mode != ORC_MODE_DEFLATE && mode != ORC_MODE_ZLIB && mode != ORC_MODE_GZIP && mode != ORC_MODE_AUTO) return NULL;
    orc_inflate *s = (orc_inflate *)malloc(sizeof *s);
    if (!s) return NULL;
    s->mode = mode;
    s->fixed_ready = 0;
    orc_inflate_reset(s);
    return s;
}

void orc_inflate_reset(orc_inflate *s)
{
    s->wrap = 0;
    s->st = s->mode < 0 ? ST_BLOCK : ST_HEAD;
    s->hold = 0;
    s->bits = 0;
    s->last = 0;
    s->gz_flags = s->gz_idx = s->gz_xlen = 0;
    s->hcrc = 0;
    s->check = 0;
    s->total_out = 0;
    s->stored_left = 0;
    s->have = 0;
    s->pend_len = s->pend_dist = 0;
    s->pend_lit = -1;
    s->wnext = s->whave = 0;
    s->msg = NULL;
    s->bad_code = Z_DATA_ERROR;
}

void orc_inflate_free(orc_inflate *s) { free(s); }
const char *orc_inflate_msg(const orc_inflate *s) { return s->msg; }

const char *orc_zlib_strerror(int32_t code)
{
    /* zlib's z_errmsg table indexed by 2 - code (zError) */
    static const char *const tab[] = {"need dictionary", "stream end", "", "file error", "stream error",
                                      "data error", "insufficient memory", "buffer error", "incompatible version", ""};
    int idx = 2 - code;
    if (idx < 0 || idx > 9) idx = 9;
    return tab[idx];
}

static void fixed_tables(orc_inflate *s)
{
    if (s->fixed_ready) return;
    uint8_t l[288];
    int i = 0;
    for (; i < 144; i++) l[i] = 8;
    for (; i < 256; i++) l[i] = 9;
    for (; i < 280; i++) l[i] = 7;
    for (; i < 288; i++) l[i] = 8;
    huff_build(&s->fixl, l, 288, LIT_ROOT, T_LENS);
    for (i = 0; i < 32; i++) l[i] = 5;
    huff_build(&s->fixd, l, 32, DIST_ROOT, T_DISTS);
    s->fixed_ready = 1;
}

static void window_update(orc_inflate *s, const uint8_t *out, size_t n)
{
    if (n >= WSIZE) {
        memcpy(s->win, out + n - WSIZE, WSIZE);
        s->wnext = 0;
        s->whave = WSIZE;
        return;
    }
    size_t first = WSIZE - s->wnext;
    if (first > n) first = n;
    memcpy(s->win + s->wnext, out, first);
    if (n > first) memcpy(s->win, out + first, n - first);
    s->wnext = (unsigned)((s->wnext + n) % WSIZE);
    s->whave = s->whave + n > WSIZE ? WSIZE : (unsigned)(s->whave + n);
}

/* One inflate(Z_NO_FLUSH) step; returns a zlib return code. */
static int inflate_step(orc_inflate *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                        size_t *in_used, size_t *out_used)
{
    const uint8_t *ip = in, *const ie = in + in_len;
    uint8_t *op = out, *const oe = out + out_len;
    const uint8_t *checked = out; /* output bytes before this are folded into s->check */
    int starved = 0;              /* left because the input ran dry inside an atomic unit */

#define REFILL() \
    while (s->bits <= 56 && ip < ie) { s->hold |= (uint64_t)(*ip++) << s->bits; s->bits += 8; }
#define NEED(n) \
    do { if (s->bits < (int)(n)) { REFILL(); if (s->bits < (int)(n)) { starved = 1; goto leave; } } } while (0)
#define PEEK(n) ((unsigned)(s->hold & ((1ull << (n)) - 1)))
#define DROP(n) (s->hold >>= (n), s->bits -= (int)(n))
#define FAIL(m) do { s->msg = (m); s->st = ST_BAD; goto leave; } while (0)
#define UPDATE_CHECK() \
    do { if (s->wrap && op > checked) { \
            s->check = s->wrap == 2 ? orc_crc32(s->check, checked, (size_t)(op - checked)) \
                                    : orc_adler32(s->check, checked, (size_t)(op - checked)); } \
         checked = op; } while (0)
#define GZ_BYTE(var) \
    do { NEED(8); (var) = PEEK(8); DROP(8); \
         if (s->gz_flags & 0x0200) { uint8_t b_ = (uint8_t)(var); s->hcrc = orc_crc32(s->hcrc, &b_, 1); } } while (0)

    for (;;) {
        switch (s->st) {
        case ST_HEAD: {
            NEED(16);
            unsigned h = PEEK(16);
            int allow_gzip = s->mode == ORC_MODE_GZIP || s->mode == ORC_MODE_AUTO;
            int allow_zlib = s->mode == ORC_MODE_ZLIB || s->mode == ORC_MODE_AUTO;
            if (allow_gzip && h == 0x8b1f) { /* gzip magic, RFC 1952 sec. 2.3.1 */
                s->wrap = 2;
                s->check = 0;
                s->hcrc = 0;
                s->gz_idx = 0;
                s->gz_flags = 0;
                s->st = ST_GZ_FIXED;
                break;
            }
            if (!allow_zlib) FAIL("incorrect header check"); /* gzip-only decoder and no gzip magic */
            if ((((h & 0xff) << 8) + (h >> 8)) % 31) FAIL("incorrect header check");
            if ((h & 0x0f) != 8) FAIL("unknown compression method");
            if (((h >> 4) & 0x0f) + 8 > 15) FAIL("invalid window size");
            DROP(16);
            s->wrap = 1;
            s->check = 1; /* adler32 seed */
            s->st = (h & 0x2000) ? ST_ZDICT : ST_BLOCK; /* FDICT, RFC 1950 sec. 2.2 */
            break;
        }
        case ST_ZDICT:
            NEED(32);
            DROP(32);
            s->st = ST_DICT;
            break;
        case ST_DICT:
            /* compu never supplies a dictionary: zlib keeps answering Z_NEED_DICT */
            s->bad_code = Z_NEED_DICT;
            goto leave;
        case ST_GZ_FIXED: {
            /* ID1 ID2 CM FLG MTIME(4) XFL OS, RFC 1952 sec. 2.3 */
            while (s->gz_idx < 10) {
                unsigned b;
                if (s->gz_idx == 4 && (s->gz_flags & 0x0200)) {
                    /* FHCRC covers the first four bytes too: fold them in once FLG is known */
                    uint8_t h4[4] = {0x1f, 0x8b, 8, (uint8_t)(s->gz_flags >> 8)};
                    s->hcrc = orc_crc32(0, h4, 4);
                }
                NEED(8);
                b = PEEK(8);
                if (s->gz_idx == 2 && b != 8) FAIL("unknown compression method");
                if (s->gz_idx == 3) {
                    if (b & 0xe0) FAIL("unknown header flags set");
                    s->gz_flags = b << 8; /* zlib keeps FLG in bits 8..15 */
                }
                DROP(8);
                if (s->gz_idx >= 4 && (s->gz_flags & 0x0200)) { uint8_t b_ = (uint8_t)b; s->hcrc = orc_crc32(s->hcrc, &b_, 1); }
                s->gz_idx++;
            }
            s->gz_idx = 0;
            s->st = (s->gz_flags & 0x0400) ? ST_GZ_EXLEN : ST_GZ_NAME;
            break;
        }
        case ST_GZ_EXLEN: {
            while (s->gz_idx < 2) {
                unsigned b;
                GZ_BYTE(b);
                s->gz_xlen = s->gz_idx ? (s->gz_xlen | (b << 8)) : b;
                s->gz_idx++;
            }
            s->st = ST_GZ_EXTRA;
            break;
        }
        case ST_GZ_EXTRA:
            while (s->gz_xlen) {
                unsigned b;
                GZ_BYTE(b);
                (void)b;
                s->gz_xlen--;
            }
            s->st = ST_GZ_NAME;
            break;
        case ST_GZ_NAME:
            if (s->gz_flags & 0x0800) {
                unsigned b;
                do { GZ_BYTE(b); } while (b);
            }
            s->st = ST_GZ_COMMENT;
            break;
        case ST_GZ_COMMENT:
            if (s->gz_flags & 0x1000) {
                unsigned b;
                do { GZ_BYTE(b); } while (b);
            }
            s->st = ST_GZ_HCRC;
            break;
        case ST_GZ_HCRC:
            if (s->gz_flags & 0x0200) {
                NEED(16);
                if (PEEK(16) != (s->hcrc & 0xffff)) FAIL("header crc mismatch");
                DROP(16);
            }
            s->check = 0; /* crc32 seed for the payload */
            s->st = ST_BLOCK;
            break;
        case ST_BLOCK: {
            if (s->last) {
                DROP(s->bits & 7);
                s->st = s->wrap ? ST_CHECK : ST_DONE;
                break;
            }
            NEED(3);
            s->last = (int)PEEK(1);
            unsigned type = (PEEK(3) >> 1);
            DROP(3);
            if (type == 0) s->st = ST_STORED_HDR;
            else if (type == 1) {
                fixed_tables(s);
                s->lith = s->fixl;
                s->dsth = s->fixd;
                s->st = ST_CODES;
            } else if (type == 2) s->st = ST_DYN_HDR;
            else FAIL("invalid block type");
            break;
        }
        case ST_STORED_HDR: {
            DROP(s->bits & 7);
            NEED(32);
            unsigned v = PEEK(16), nv = (unsigned)((s->hold >> 16) & 0xffff);
            if (v != (nv ^ 0xffff)) FAIL("invalid stored block lengths");
            DROP(32);
            s->stored_left = v;
            s->st = ST_STORED_COPY;
            break;
        }
        case ST_STORED_COPY: {
            while (s->stored_left && s->bits >= 8 && op < oe) {
                *op++ = (uint8_t)PEEK(8);
                DROP(8);
                s->stored_left--;
            }
            if (s->stored_left && s->bits == 0) {
                size_t n = s->stored_left;
                if ((size_t)(ie - ip) < n) n = (size_t)(ie - ip);
                if ((size_t)(oe - op) < n) n = (size_t)(oe - op);
                memcpy(op, ip, n);
                ip += n;
                op += n;
                s->stored_left -= (unsigned)n;
            }
            if (s->stored_left) {
                if (op < oe) starved = 1; /* room left, so it is the input that ran out */
                goto leave;
            }
            s->st = ST_BLOCK;
            break;
        }
        case ST_DYN_HDR:
            NEED(14);
            s->nlen = PEEK(5) + 257;
            DROP(5);
            s->ndist = PEEK(5) + 1;
            DROP(5);
            s->ncode = PEEK(4) + 4;
            DROP(4);
            if (s->nlen > 286 || s->ndist > 30) FAIL("too many length or distance symbols");
            s->have = 0;
            memset(s->lens, 0, 19);
            s->st = ST_DYN_CLENS;
            break;
        case ST_DYN_CLENS:
            while (s->have < s->ncode) {
                NEED(3);
                s->lens[CL_ORDER[s->have++]] = (uint8_t)PEEK(3);
                DROP(3);
            }
            if (huff_build(&s->clh, s->lens, 19, CL_ROOT, T_CODES)) FAIL("invalid code lengths set");
            s->have = 0;
            s->st = ST_DYN_LENS;
            break;
        case ST_DYN_LENS: {
            while (s->have < s->nlen + s->ndist) {
                REFILL();
                uint32_t e = huff_lookup(&s->clh, s->hold);
                unsigned cl = E_CL(e), sym = E_BASE(e);
                if ((int)cl > s->bits) { starved = 1; goto leave; }
                if (sym < 16) {
                    DROP(cl);
                    s->lens[s->have++] = (uint8_t)sym;
                    continue;
                }
                unsigned eb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                if ((int)(cl + eb) > s->bits) { starved = 1; goto leave; }
                unsigned rep, val = 0;
                if (sym == 16) {
                    if (s->have == 0) FAIL("invalid bit length repeat");
                    val = s->lens[s->have - 1];
                    rep = 3 + ((unsigned)(s->hold >> cl) & 3);
                } else if (sym == 17) rep = 3 + ((unsigned)(s->hold >> cl) & 7);
                else rep = 11 + ((unsigned)(s->hold >> cl) & 0x7f);
                if (s->have + rep > s->nlen + s->ndist) FAIL("invalid bit length repeat");
                DROP(cl + eb);
                while (rep--) s->lens[s->have++] = (uint8_t)val;
            }
            if (s->lens[256] == 0) FAIL("invalid code -- missing end-of-block");
            if (huff_build(&s->lith, s->lens, (int)s->nlen, LIT_ROOT, T_LENS)) FAIL("invalid literal/lengths set");
            if (huff_build(&s->dsth, s->lens + s->nlen, (int)s->ndist, DIST_ROOT, T_DISTS)) FAIL("invalid distances set");
            s->st = ST_CODES;
            break;
        }
        case ST_CODES: {
            for (;;) {
                /* zlib decodes the next symbol first and only then asks for output space */
                if (s->pend_lit >= 0) {
                    if (op == oe) goto leave;
                    *op++ = (uint8_t)s->pend_lit;
                    s->pend_lit = -1;
                }
                if (s->pend_len) {
                    if (op == oe) goto leave;
                    unsigned dist = s->pend_dist;
                    while (s->pend_len && op < oe) {
                        size_t produced = (size_t)(op - out);
                        if (dist > produced) { /* source predates this call: take it from the window */
                            unsigned back = dist - (unsigned)produced; /* <= whave */
                            *op++ = s->win[(s->wnext + WSIZE - back) % WSIZE];
                            s->pend_len--;
                        } else if (dist >= s->pend_len && (size_t)(oe - op) >= s->pend_len) {
                            memcpy(op, op - dist, s->pend_len);
                            op += s->pend_len;
                            s->pend_len = 0;
                        } else {
                            *op = *(op - dist);
                            op++;
                            s->pend_len--;
                        }
                    }
                    if (s->pend_len) goto leave;
                }
                REFILL();
                uint64_t h = s->hold;
                int b = s->bits;
                uint32_t e = huff_lookup(&s->lith, h);
                unsigned cl = E_CL(e), eb = E_EB(e), kind = E_KIND(e);
                if ((int)(cl + eb) > b) { starved = 1; goto leave; }
                if (kind == K_LIT) {
                    DROP(cl);
                    if (op < oe) *op++ = (uint8_t)E_BASE(e);
                    else s->pend_lit = (int)E_BASE(e);
                    continue;
                }
                if (kind == K_EOB) {
                    DROP(cl);
                    break;
                }
                if (kind == K_BAD) FAIL("invalid literal/length code");
                unsigned len = E_BASE(e) + ((unsigned)(h >> cl) & ((1u << eb) - 1));
                unsigned n1 = cl + eb;
                h >>= n1;
                b -= (int)n1;
                uint32_t e2 = huff_lookup(&s->dsth, h);
                unsigned cl2 = E_CL(e2), eb2 = E_EB(e2);
                if ((int)(cl2 + eb2) > b) { starved = 1; goto leave; }
                if (E_KIND(e2) == K_BAD) {
                    DROP(n1); /* zlib has already dropped the length code when it meets the bad distance */
                    FAIL("invalid distance code");
                }
                unsigned dist = E_BASE(e2) + ((unsigned)(h >> cl2) & ((1u << eb2) - 1));
                DROP(n1 + cl2 + eb2);
                if ((uint64_t)dist > s->total_out + (uint64_t)(op - out)) FAIL("invalid distance too far back");
                s->pend_len = len;
                s->pend_dist = dist;
            }
            s->st = ST_BLOCK;
            break;
        }
        case ST_CHECK: {
            UPDATE_CHECK();
            NEED(32);
            uint32_t v = (uint32_t)PEEK(32);
            if (s->wrap == 1) v = (v >> 24) | ((v >> 8) & 0xff00) | ((v << 8) & 0xff0000) | (v << 24); /* big-endian adler */
            if (v != s->check) FAIL("incorrect data check");
            DROP(32);
            s->st = s->wrap == 2 ? ST_LENGTH : ST_DONE;
            break;
        }
        case ST_LENGTH: {
            NEED(32);
            uint32_t total = (uint32_t)(s->total_out + (uint64_t)(op - out));
            if ((uint32_t)PEEK(32) != total) FAIL("incorrect length check");
            DROP(32);
            s->st = ST_DONE;
            break;
        }
        case ST_DONE:
        case ST_BAD:
            goto leave;
        }
    }
leave:
    if (!starved) {
        /* hand whole unread bytes pulled during this call back to the caller */
        while (s->bits >= 8 && ip > in) {
            ip--;
            s->bits -= 8;
        }
        s->hold &= s->bits ? ((1ull << s->bits) - 1) : 0;
    }
    UPDATE_CHECK();
    if (op > out) window_update(s, out, (size_t)(op - out));
    s->total_out += (uint64_t)(op - out);
    *in_used = (size_t)(ip - in);
    *out_used = (size_t)(op - out);
    if (s->st == ST_DONE) return Z_STREAM_END;
    if (s->st == ST_BAD || s->st == ST_DICT) return s->bad_code;
    if (*in_used == 0 && *out_used == 0) return Z_BUF_ERROR;
    return Z_OK;
}

/* internal_zlib_impl_decode!, src/decoder/mod.rs:459-486 */
orc_decode_t orc_inflate_decode(orc_inflate *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    size_t iu = 0, ou = 0;
    int rc = inflate_step(s, in, in_len, out, out_len, &iu, &ou);
    orc_decode_t r;
    r.input_remain = in_len - iu;
    r.output_remain = out_len - ou;
    r.err = 0;
    if (rc == Z_OK) r.status = r.input_remain == 0 ? ORC_NEED_INPUT : ORC_NEED_OUTPUT; /* mod.rs:476-479 */
    else if (rc == Z_STREAM_END) r.status = ORC_FINISHED;                               /* mod.rs:480 */
    else if (rc == Z_BUF_ERROR) r.status = ORC_NEED_OUTPUT;                             /* mod.rs:481 */
    else {
        r.status = -1;
        r.err = rc; /* mod.rs:482 */
    }
    return r;
}

/* ---- many independent units (CPU baseline loop) ---- */
typedef struct {
    int mode;
    size_t lo, hi;
    const uint8_t *in_base;
    const uint64_t *in_off;
    const uint32_t *in_len;
    uint8_t *out_base;
    const uint64_t *out_off;
    const uint32_t *out_cap;
    uint32_t *out_len;
    int32_t *status;
    size_t bad;
} unit_job;

static void *unit_worker(void *arg)
{
    unit_job *j = (unit_job *)arg;
    orc_inflate *s = orc_inflate_new(j->mode); /* one decoder per worker, zlib_ng.rs:61 */
    j->bad = 0;
    for (size_t i = j->lo; i < j->hi; i++) {
        orc_decode_t r = orc_inflate_decode(s, j->in_base + j->in_off[i], j->in_len[i], j->out_base + j->out_off[i], j->out_cap[i]);
        j->out_len[i] = (uint32_t)(j->out_cap[i] - r.output_remain);
        j->status[i] = r.err ? r.err : r.status;
        if (r.err || r.status != ORC_FINISHED) j->bad++;
        orc_inflate_reset(s); /* zlib_ng.rs:99-108 */
    }
    orc_inflate_free(s);
    return NULL;
}

size_t orc_inflate_units(int mode, size_t n, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len,
                         uint8_t *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                         int32_t *status, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n && n > 0) n_threads = (int)n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    unit_job *jobs = (unit_job *)malloc(sizeof(unit_job) * (size_t)n_threads);
    size_t bad = 0;
    for (int t = 0; t < n_threads; t++) {
        unit_job *j = &jobs[t];
        j->mode = mode;
        j->lo = n * (size_t)t / (size_t)n_threads;
        j->hi = n * (size_t)(t + 1) / (size_t)n_threads;
        j->in_base = in_base; j->in_off = in_off; j->in_len = in_len;
        j->out_base = out_base; j->out_off = out_off; j->out_cap = out_cap;
        j->out_len = out_len; j->status = status;
        if (n_threads == 1) unit_worker(j);
        else pthread_create(&th[t], NULL, unit_worker, j);
    }
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        bad += jobs[t].bad;
    }
    free(th);
    free(jobs);
    return bad;
}
