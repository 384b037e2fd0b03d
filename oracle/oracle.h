/*
 * oracle/oracle.h -- CPU restatement of the codec arithmetic behind compu's
 * zlib-ng / zstd backends.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (compu_amd/, include/) never links or calls it.
 *
 * What it restates (the arithmetic itself lives in un-vendored third-party C
 * that is absent from /root/reference: libz-ng-sys ^1.1.9 -> zlib-ng 2.x,
 * zstd-sys ^2.0.8 -> zstd >= 1.5.5; versions unpinned, Cargo.lock is git-ignored):
 *   - inflate(Z_NO_FLUSH) as called at  src/decoder/mod.rs:470   (RFC 1951/1950/1952)
 *   - the status mapping of             src/decoder/mod.rs:475-483
 *   - ZSTD_decompressStream as called   src/decoder/zstd.rs:110  (RFC 8878)
 *   - the status mapping of             src/decoder/zstd.rs:113-135
 *   - deflate(level 1) as called at     src/encoder/mod.rs:352   (format-valid; bytes unpinned)
 *
 * Parity pin: the reference's own fixtures tests/data/{10x10y,alice29.txt}.compressed.{gz,zstd}
 * (copied as data into tests/golden/) through the phases of tests/decoder.rs:21-77,
 * plus cross-checks against system zlib 1.2.11 / libzstd in the build container
 * (tests/test_oracle_*.py).
 */
#ifndef COMPU_ORACLE_H
#define COMPU_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DecodeStatus, src/decoder/mod.rs:139-146 (enum order preserved) */
enum { ORC_NEED_INPUT = 0, ORC_NEED_OUTPUT = 1, ORC_FINISHED = 2 };

/* Decode, src/decoder/mod.rs:150-157: status is Ok(DecodeStatus) when err == 0,
 * Err(DecodeError(err)) otherwise. */
typedef struct {
    size_t input_remain;
    size_t output_remain;
    int32_t status;
    int32_t err;
} orc_decode_t;

/* ZlibMode, src/decoder/zlib_common.rs:4-15 (windowBits values) */
enum { ORC_MODE_DEFLATE = -15, ORC_MODE_ZLIB = 15, ORC_MODE_GZIP = 31, ORC_MODE_AUTO = 47 };

typedef struct orc_inflate orc_inflate;

/* Interface::zlib_ng(mode), src/decoder/zlib_ng.rs:61-90 */
orc_inflate *orc_inflate_new(int mode);
/* decode_fn, src/decoder/zlib_ng.rs:94-96 + internal_zlib_impl_decode! src/decoder/mod.rs:459-486 */
orc_decode_t orc_inflate_decode(orc_inflate *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
/* reset_fn, src/decoder/zlib_ng.rs:99-108 */
void orc_inflate_reset(orc_inflate *s);
/* drop_fn, src/decoder/zlib_ng.rs:111-115 */
void orc_inflate_free(orc_inflate *s);
/* last zlib-style "msg" (diagnostic only, not part of the compu contract) */
const char *orc_inflate_msg(const orc_inflate *s);
/* describe_error_fn, src/decoder/zlib_ng.rs:118-123 (zError table) */
const char *orc_zlib_strerror(int32_t code);

uint32_t orc_crc32(uint32_t crc, const uint8_t *p, size_t n);   /* RFC 1952 sec. 8 */
uint32_t orc_adler32(uint32_t adler, const uint8_t *p, size_t n); /* RFC 1950 sec. 9 */

/* One-shot helper over many independent units (the CPU baseline loop of
 * one decoder per worker, reset per unit).  Returns the
 * number of units that did not finish cleanly. */
size_t orc_inflate_units(int mode, size_t n, const uint8_t *in_base, const uint64_t *in_off,
                         const uint32_t *in_len, uint8_t *out_base, const uint64_t *out_off,
                         const uint32_t *out_cap, uint32_t *out_len, int32_t *status,
                         int n_threads);

/* ---- zstd frame decode (oracle_zstd.c) ---- */
typedef struct orc_zstd orc_zstd;
/* Interface::zstd(opts), src/decoder/zstd.rs:81-94; window_log_max 0 = default (27) */
orc_zstd *orc_zstd_new(int window_log_max);
/* decode_fn, src/decoder/zstd.rs:98-136 */
orc_decode_t orc_zstd_decode(orc_zstd *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
/* reset_fn, src/decoder/zstd.rs:139-148 */
void orc_zstd_reset(orc_zstd *s);
void orc_zstd_free(orc_zstd *s);
/* describe_error_fn, src/decoder/zstd.rs:159-164 */
const char *orc_zstd_strerror(int32_t code);
uint64_t orc_xxh64(const uint8_t *p, size_t n, uint64_t seed);
size_t orc_zstd_units(size_t n, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len,
                      uint8_t *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                      uint32_t *out_len, int32_t *status, int n_threads);

/* ---- deflate level-1 style encoder (oracle_deflate.c) ---- */
/* EncodeOp src/encoder/mod.rs:12-23; EncodeStatus src/encoder/mod.rs:27-38 */
enum { ORC_OP_PROCESS = 0, ORC_OP_FLUSH = 1, ORC_OP_FINISH = 2 };
enum { ORC_ENC_CONTINUE = 0, ORC_ENC_NEED_OUTPUT = 1, ORC_ENC_FINISHED = 2, ORC_ENC_ERROR = 3 };
typedef struct {
    size_t input_remain;
    size_t output_remain;
    int32_t status;
} orc_encode_t;
typedef struct orc_deflate orc_deflate;
/* Interface::zlib_ng(opts), src/encoder/zlib_ng.rs:50-87; mode = -15 | 15 | 31 */
orc_deflate *orc_deflate_new(int mode, int level);
int orc_deflate_set_strategy(orc_deflate *s, int strategy); /* ZlibStrategy 0..4, default 0 */
/* encode_fn, src/encoder/zlib_ng.rs:90-92 + internal_zlib_impl_encode! src/encoder/mod.rs:334-370 */
orc_encode_t orc_deflate_encode(orc_deflate *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int op);
void orc_deflate_reset(orc_deflate *s);
void orc_deflate_free(orc_deflate *s);

#ifdef __cplusplus
}
#endif
#endif
