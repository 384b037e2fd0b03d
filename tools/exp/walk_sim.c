// Walk-strategy simulator (CPU, exploration only; not product, not oracle).
// Reads raw-deflate units (file: u32 n, then per unit u32 len + bytes), decodes every block truly, and for every bit
// position of a block's token area memoizes what a speculative decoder starting there would see (token bits, kind).
// Then replays walk policies and prints wave-steps per unit.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint16_t sym[320]; uint16_t count[16]; } Huff;
static const uint8_t *g_in; static uint32_t g_bits;
static inline uint32_t peek(uint32_t pos, int n) { // n<=25
    uint32_t b = pos >> 3; uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)( (b + k) * 8 < g_bits + 64 ? g_in[b + k] : 0) << (8 * k);
    return (uint32_t)((v >> (pos & 7)) & ((1u << n) - 1));
}
static void build(Huff *h, const uint8_t *len, int n) {
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++) h->count[len[i]]++;
    h->count[0] = 0; uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h->count[l];
    for (int i = 0; i < n; i++) if (len[i]) h->sym[offs[len[i]]++] = i;
}
// returns symbol or -1; *pos advanced
static int dec(const Huff *h, uint32_t *pos) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        code |= peek(*pos, 1); (*pos)++;
        int c = h->count[l];
        if (code - c < first) return h->sym[index + (code - first)];
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}
static int g_cur_cl, g_cur_dcl;
static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint8_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
// token at pos: returns bits consumed (0 = halt: EOB/invalid), *islit
static int token(const Huff *hl, const Huff *hd, uint32_t pos, int *islit, int *litlen) {
    uint32_t p = pos; int s = dec(hl, &p);
    if (s < 0 || s == 256 || s >= 286) return 0;
    if (s < 256) { g_cur_cl = p - pos; *islit = 1; *litlen = p - pos; return p - pos; }
    g_cur_cl = p - pos; *islit = 0; p += lext[s - 257]; uint32_t p0_ = p;
    int d = dec(hd, &p); if (d < 0 || d >= 30) return 0;
    g_cur_dcl = p - p0_; p += dext[d]; return p - pos;
}
typedef struct { uint32_t start, end; uint8_t *tb; uint8_t *lit; uint32_t ntok; } Block; // tb[pos-start]

static int g_cut=0, g_nopause=0; static double g_steps_cut, g_lanetok_cut; static double g_steps_cur, g_steps_roll, g_units; static double g_tok, g_lanetok_cur, g_lanetok_roll;
static uint64_t g_pairs9, g_pairs10, g_pairs11, g_pairs12, g_lits, g_litpairs_possible;
static uint64_t g_dhist[4096]; static uint64_t g_clh[16], g_dclh[16], g_ntokh;

#define MAXSEG 4096
static void sim_block(Block *b, int S, int XT, int ring, int refill_every, int refill_min) {
    uint32_t n = b->end - b->start;
    // ---- true chain marks
    // ---- current scheme: super-rounds of 64 segments from true position B
    {
        uint32_t B = 0; double steps = 0, lanetok = 0;
        static uint8_t mark[1 << 22]; 
        while (B < n && b->tb[B]) {
            // simulate in lockstep
            uint32_t p[64], lim[64]; int act[64]; int nact = 0;
            memset(mark + B, 0, (size_t)64 * S + 64 < n - B ? (size_t)64 * S + 64 : n - B);
            for (int l = 0; l < 64; l++) { p[l] = B + l * S; lim[l] = p[l] + S + XT; if (lim[l] > B + 64u * S) lim[l] = B + 64u * S; act[l] = p[l] < n; nact += act[l]; }
            int st = 0; uint32_t endpos = 0; int joined_to[64]; uint32_t stop[64]; int cnt[64];
            for (int l = 0; l < 64; l++) { joined_to[l] = -1; stop[l] = 0; cnt[l] = 0; }
            while (nact) {
                st++;
                for (int sub = 0; sub < 1; sub++)
                for (int l = 0; l < 64; l++) if (act[l]) {
                    uint32_t seg = (p[l] - B) / S; if (seg > 63) seg = 63;
                    if ((int)seg == l) mark[p[l]] = 1;
                    else if (mark[p[l]]) { act[l] = 0; nact--; joined_to[l] = seg; stop[l] = p[l]; continue; }
                    int t = p[l] < n ? b->tb[p[l]] : 0;
                    if (!t) { act[l] = 0; nact--; stop[l] = p[l]; continue; }
                    p[l] += t; cnt[l]++; lanetok++;
                    if (p[l] >= lim[l] || cnt[l] >= 192) { act[l] = 0; nact--; stop[l] = p[l]; }
                }
            }
            steps += st;
            // follow chain from lane 0
            int l = 0; for (;;) { if (joined_to[l] < 0) { endpos = stop[l]; break; } l = joined_to[l]; }
            if (endpos <= B) break;
            B = endpos;
        }
        g_steps_cur += steps; g_lanetok_cur += lanetok;
    }
    // ---- early-cut scheme: super-rounds of 64 segments from true position B
    {
        uint32_t B = 0; double steps = 0, lanetok = 0;
        static uint8_t mark[1 << 22]; 
        while (B < n && b->tb[B]) {
            // simulate in lockstep
            uint32_t p[64], lim[64]; int act[64]; int nact = 0;
            memset(mark + B, 0, (size_t)64 * S + 64 < n - B ? (size_t)64 * S + 64 : n - B);
            for (int l = 0; l < 64; l++) { p[l] = B + l * S; lim[l] = p[l] + S + XT; if (lim[l] > B + 64u * S) lim[l] = B + 64u * S; act[l] = p[l] < n; nact += act[l]; }
            int st = 0; uint32_t endpos = 0; int joined_to[64]; uint32_t stop[64]; int cnt[64];
            for (int l = 0; l < 64; l++) { joined_to[l] = -1; stop[l] = 0; cnt[l] = 0; }
            while (nact > g_cut) {
                st++;
                for (int sub = 0; sub < 1; sub++)
                for (int l = 0; l < 64; l++) if (act[l]) {
                    uint32_t seg = (p[l] - B) / S; if (seg > 63) seg = 63;
                    if ((int)seg == l) mark[p[l]] = 1;
                    else if (mark[p[l]]) { act[l] = 0; nact--; joined_to[l] = seg; stop[l] = p[l]; continue; }
                    int t = p[l] < n ? b->tb[p[l]] : 0;
                    if (!t) { act[l] = 0; nact--; stop[l] = p[l]; continue; }
                    p[l] += t; cnt[l]++; lanetok++;
                    if (p[l] >= lim[l] || cnt[l] >= 192) { act[l] = 0; nact--; stop[l] = p[l]; }
                }
            }
            steps += st;
            for (int l2 = 0; l2 < 64; l2++) if (act[l2]) { stop[l2] = p[l2]; joined_to[l2] = -1; }
            // follow chain from lane 0
            int l = 0; for (;;) { if (joined_to[l] < 0) { endpos = stop[l]; break; } l = joined_to[l]; }
            if (endpos <= B) break;
            B = endpos;
        }
        g_steps_cut += steps; g_lanetok_cut += lanetok;
    }
    // ---- sync distance histogram: for each grid segment start s_k (k>=1) how many tokens until on the true chain
    {
        static uint8_t truem[1 << 22]; memset(truem, 0, n + 64);
        uint32_t p = 0; while (p < n && b->tb[p]) { truem[p] = 1; p += b->tb[p]; } truem[p] = 1;
        for (uint32_t s = S; s < n; s += S) { uint32_t q = s; int d = 0; while (q < n && !truem[q] && b->tb[q] && d < 4095) { q += b->tb[q]; d++; } g_dhist[d]++; }
    }
    // ---- rolling scheme: fixed grid over the block, lanes claim next segment; pause rule; ring of `ring` segments
    {
        uint32_t nseg = (n + S - 1) / S; if (nseg > MAXSEG) nseg = MAXSEG;
        static uint8_t mark[1 << 22]; memset(mark, 0, n + 64);
        static uint8_t owner_done[MAXSEG]; memset(owner_done, 0, sizeof owner_done);
        static uint8_t seg_joined[MAXSEG]; memset(seg_joined, 0, sizeof seg_joined);
        int lane_seg[64]; uint32_t p[64]; int cnt[64]; uint32_t lim[64];
        for (int l = 0; l < 64; l++) lane_seg[l] = -1;
        uint32_t next_seg = 0, lo = 0; double steps = 0, lanetok = 0; int since = 0;
        for (;;) {
            // refill policy
            int idle = 0; for (int l = 0; l < 64; l++) if (lane_seg[l] < 0) idle++;
            if (getenv("LOMINP")) { uint32_t mp = 0xffffffffu; for (int l = 0; l < 64; l++) if (lane_seg[l] >= 0 && p[l] < mp) mp = p[l]; lo = mp == 0xffffffffu ? next_seg : mp / S; } else while (lo < nseg && seg_joined[lo]) lo++;
            if (idle == 64 && next_seg >= nseg) break;
            if (idle && next_seg < nseg && (idle >= refill_min || since >= refill_every || idle == 64)) {
                for (int l = 0; l < 64; l++) if (lane_seg[l] < 0 && next_seg < nseg && next_seg < lo + ring) {
                    lane_seg[l] = next_seg; p[l] = next_seg * S; cnt[l] = 0; lim[l] = p[l] + S + XT; next_seg++;
                }
                since = 0;
            }
            int any = 0; for (int l = 0; l < 64; l++) if (lane_seg[l] >= 0) any = 1;
            if (!any) { if (next_seg >= nseg) break; /* ring full with no active lanes cannot happen */ break; }
            steps++; since++;
            for (int l = 0; l < 64; l++) if (lane_seg[l] >= 0) {
                int k = lane_seg[l]; uint32_t seg = p[l] / S;
                int fin = 0;
                if ((int)seg == k) mark[p[l]] = 1;
                else {
                    if (!owner_done[k]) owner_done[k] = 1;
                    if (!g_nopause && (seg >= next_seg || !owner_done[seg])) { if (seg < nseg) continue; /* pause */ }
                    if (mark[p[l]]) fin = 1;
                }
                if (!fin) {
                    int t = p[l] < n ? b->tb[p[l]] : 0;
                    if (!t) fin = 1; else { p[l] += t; cnt[l]++; lanetok++; if (p[l] / S != (uint32_t)k) owner_done[k] = 1; if (p[l] >= lim[l] || cnt[l] >= 192) fin = 1; }
                }
                if (fin) { owner_done[k] = 1; seg_joined[k] = 1; lane_seg[l] = -1; }
            }
        }
        g_steps_roll += steps; g_lanetok_roll += lanetok;
    }
}

int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); int S = argc > 2 ? atoi(argv[2]) : 288, XT = argc > 3 ? atoi(argv[3]) : 1024;
    g_cut = getenv("CUT") ? atoi(getenv("CUT")) : 0; g_nopause = getenv("NOPAUSE") != 0; int ring = argc > 4 ? atoi(argv[4]) : 96, re = argc > 5 ? atoi(argv[5]) : 1, rmin = argc > 6 ? atoi(argv[6]) : 1;
    uint32_t nu; fread(&nu, 4, 1, f);
    for (uint32_t u = 0; u < nu; u++) {
        uint32_t len; fread(&len, 4, 1, f); uint8_t *in = calloc(len + 64, 1); fread(in, 1, len, f);
        g_in = in; g_bits = len * 8; uint32_t pos = 0; int last = 0;
        while (!last) {
            last = peek(pos, 1); int type = peek(pos + 1, 2); pos += 3;
            if (type != 2 && type != 1) { fprintf(stderr, "type %d unsupported\n", type); return 1; }
            uint8_t lens[320]; memset(lens, 0, sizeof lens); Huff hl, hd; int nlen = 288, ndist = 30;
            if (type == 1) { for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8; build(&hl, lens, 288); uint8_t dl[30]; memset(dl, 5, 30); build(&hd, dl, 30); }
            else {
                nlen = peek(pos, 5) + 257; ndist = peek(pos + 5, 5) + 1; int ncode = peek(pos + 10, 4) + 4; pos += 14;
                static const uint8_t ord[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                uint8_t cl[19]; memset(cl, 0, 19); for (int i = 0; i < ncode; i++) { cl[ord[i]] = peek(pos, 3); pos += 3; }
                Huff hc; build(&hc, cl, 19); int i = 0;
                while (i < nlen + ndist) { int s = dec(&hc, &pos); if (s < 16) lens[i++] = s; else if (s == 16) { int r = 3 + peek(pos, 2); pos += 2; while (r--) { lens[i] = lens[i - 1]; i++; } } else if (s == 17) { int r = 3 + peek(pos, 3); pos += 3; while (r--) lens[i++] = 0; } else { int r = 11 + peek(pos, 7); pos += 7; while (r--) lens[i++] = 0; } }
                build(&hl, lens, nlen); build(&hd, lens + nlen, ndist);
            }
            // true decode to find block end
            uint32_t p = pos; uint32_t ntok = 0; int prevlit = 0, prevlen = 0;
            for (;;) { int il = 0, ll = 0; int t = token(&hl, &hd, p, &il, &ll); if (!t) break; g_clh[g_cur_cl]++; g_ntokh++; if (!il) g_dclh[g_cur_dcl]++; if (il) { g_lits++; if (prevlit) { int s = prevlen + ll; g_litpairs_possible++; if (s <= 9) g_pairs9++; if (s <= 10) g_pairs10++; if (s <= 11) g_pairs11++; if (s <= 12) g_pairs12++; } } prevlit = il; prevlen = ll; p += t; ntok++; }
            uint32_t eob = p; { uint32_t q = p; dec(&hl, &q); p = q; }
            Block b; b.start = pos; b.end = eob; b.ntok = ntok; uint32_t n = eob - pos;
            b.tb = calloc(n + 64, 1);
            for (uint32_t q = 0; q < n; q++) { int il, ll; int t = token(&hl, &hd, pos + q, &il, &ll); if (pos + q + t > eob + 0 && t) { /* runs past EOB start: allow */ } b.tb[q] = t; }
            g_tok += ntok;
            sim_block(&b, S, XT, ring, re, rmin);
            free(b.tb); pos = p;
        }
        g_units++; free(in);
    }
    printf("units %.0f tokens/unit %.0f | current: steps/unit %.0f lane-tokens/unit %.0f (x%.2f) | rolling(S=%d ring=%d every=%d min=%d): steps/unit %.0f lane-tokens %.0f (x%.2f)\n",
           g_units, g_tok / g_units, g_steps_cur / g_units, g_lanetok_cur / g_units, g_lanetok_cur / g_tok, S, ring, re, rmin, g_steps_roll / g_units, g_lanetok_roll / g_units, g_lanetok_roll / g_tok);
    printf("early-cut(active<=%d): steps/unit %.0f lane-tokens %.0f\n", g_cut, g_steps_cut / g_units, g_lanetok_cut / g_units);
    printf("literals/unit %.0f; adjacent literal pairs with total code bits <=9: %.3f <=10: %.3f <=11: %.3f <=12: %.3f (of adjacent lit-lit pairs %.0f/unit)\n", g_lits / g_units,
           (double)g_pairs9 / g_litpairs_possible, (double)g_pairs10 / g_litpairs_possible, (double)g_pairs11 / g_litpairs_possible, (double)g_pairs12 / g_litpairs_possible, g_litpairs_possible / g_units);
    printf("lit/len code length hist (per token):"); for (int i=1;i<16;i++) printf(" %d:%.3f", i, (double)g_clh[i]/g_ntokh); printf("\ndist code length hist (per match):"); { uint64_t m=0; for (int i=0;i<16;i++) m+=g_dclh[i]; for (int i=1;i<16;i++) printf(" %d:%.3f", i, (double)g_dclh[i]/m);} printf("\n");
    uint64_t tot = 0, acc = 0; for (int d = 0; d < 4096; d++) tot += g_dhist[d]; double mean = 0; for (int d = 0; d < 4096; d++) mean += (double)d * g_dhist[d] / tot;
    printf("sync distance (tokens) mean %.1f; quantiles:", mean); double qs[] = {0.5, 0.75, 0.9, 0.95, 0.98, 0.99, 0.999}; int qi = 0;
    for (int d = 0; d < 4096 && qi < 7; d++) { acc += g_dhist[d]; while (qi < 7 && (double)acc / tot >= qs[qi]) { printf(" p%.1f=%d", qs[qi] * 100, d); qi++; } }
    printf("\n");
    return 0;
}
