/*
 * bench_support/synth.c -- deterministic synthetic payloads for the batched-decode benchmark
 * (SURVEY.md sec. 8d).  Not part of the product and not part of the oracle: it only makes inputs.
 *
 * Unit i is generated from seed splitmix64(0xC0FFEE00D15EA5E ^ i) by xoshiro256** driving a
 * two-source model: with probability p_match emit a back-reference (length 3 + geometric, capped
 * at 258; distance uniform in [1, min(pos, 32768)]), otherwise a literal drawn from a Zipf-like
 * distribution over 256 byte values with exponent zipf_s.  (p_match, zipf_s) are calibrated so
 * that zlib level 6 compresses a 64 KiB unit to 0.50 +- 0.02 of its size.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
typedef struct { uint64_t s[4]; } xo_t;
static inline uint64_t xo_next(xo_t *g)
{
    uint64_t *s = g->s, r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
}

typedef struct {
    uint8_t *out;
    uint64_t first_unit;
    size_t lo, hi, unit_size;
    double p_match, zipf_s, len_p;
} job_t;

static void gen_unit(uint8_t *dst, size_t n, uint64_t unit, const uint8_t *lit_tab, const uint8_t *len_tab, uint32_t pm_thresh)
{
    uint64_t sm = 0xC0FFEE00D15EA5Eull ^ unit;
    xo_t g;
    for (int k = 0; k < 4; k++) g.s[k] = splitmix64(&sm);
    size_t pos = 0;
    while (pos < n) {
        uint64_t r = xo_next(&g);
        if (pos > 0 && (uint32_t)r < pm_thresh) {
            size_t len = 3 + (size_t)len_tab[(r >> 32) & 0xffff];
            if (len > 258) len = 258;
            if (len > n - pos) len = n - pos;
            uint32_t maxd = pos < 32768 ? (uint32_t)pos : 32768u;
            size_t dist = 1 + (size_t)(((r >> 48) * maxd) >> 16); /* uniform in [1, maxd] */
            for (size_t k = 0; k < len; k++) dst[pos + k] = dst[pos + k - dist];
            pos += len;
        } else {
            dst[pos++] = lit_tab[r >> 48];
        }
    }
}

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    /* inverse-CDF tables with 16-bit resolution */
    uint8_t *lit_tab = malloc(65536), *len_tab = malloc(65536);
    double w[256], tot = 0, acc = 0;
    for (int k = 0; k < 256; k++) { w[k] = pow(k + 1.0, -j->zipf_s); tot += w[k]; }
    int sym = 0;
    acc = w[0] / tot;
    for (int v = 0; v < 65536; v++) {
        while (sym < 255 && (v + 0.5) / 65536.0 >= acc) { sym++; acc += w[sym] / tot; }
        lit_tab[v] = (uint8_t)((sym * 167 + 13) & 255);
    }
    int g = 0;
    double q = 1.0 - j->len_p, cdf = j->len_p; /* P(G = g) = len_p * q^g */
    for (int v = 0; v < 65536; v++) {
        while (g < 255 && (v + 0.5) / 65536.0 >= cdf) { g++; cdf += j->len_p * pow(q, g); }
        len_tab[v] = (uint8_t)g;
    }
    uint32_t pm = (uint32_t)(j->p_match * 4294967295.0);
    for (size_t i = j->lo; i < j->hi; i++) gen_unit(j->out + i * j->unit_size, j->unit_size, j->first_unit + i, lit_tab, len_tab, pm);
    free(lit_tab);
    free(len_tab);
    return NULL;
}

/* Fill out[0 .. n_units*unit_size) with units first_unit .. first_unit+n_units-1. */
void synth_units(uint8_t *out, uint64_t first_unit, size_t n_units, size_t unit_size, double p_match, double zipf_s,
                 double len_p, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_units && n_units) n_threads = (int)n_units;
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)n_threads);
    job_t *jobs = malloc(sizeof(job_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (job_t){out, first_unit, n_units * (size_t)t / (size_t)n_threads, n_units * (size_t)(t + 1) / (size_t)n_threads,
                          unit_size, p_match, zipf_s, len_p};
        if (n_threads == 1) worker(&jobs[t]);
        else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (n_threads > 1) for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}
