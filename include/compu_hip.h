/*
 * compu_hip.h -- C ABI of the MI355X (gfx950) batched compression backend for compu.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.  Each entry point
 * names the reference interface it replaces (paths relative to the compu crate root).  A Rust
 * `hip` Interface variant binds these 1:1 (INTEGRATION.md shows the glue).
 *
 * Everything here runs on the GPU.  There is no CPU codec behind this library: if no HIP device
 * is usable the constructors return NULL / the batch calls return CHIP_E_NO_DEVICE.
 */
#ifndef COMPU_HIP_H
#define COMPU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- shared enums -------------------------------------------------------------------------- */

/* decoder::DecodeStatus, src/decoder/mod.rs:139-146 (same order) */
enum { CHIP_NEED_INPUT = 0, CHIP_NEED_OUTPUT = 1, CHIP_FINISHED = 2 };
/* Batch status only: the zlib header asks for a preset dictionary.  zlib answers Z_NEED_DICT (+2),
 * which compu passes through as Err(DecodeError(2)) (src/decoder/mod.rs:482); the value 2 is taken
 * by CHIP_FINISHED in the per-unit status array, hence a code of its own.  chip_decode() reports
 * it as err = 2 like the reference. */
enum { CHIP_NEED_DICT = 3 };
/* Batch status only, CHIP_FMT_DETECT: the unit is neither gzip, zlib nor zstd (Detection::Unknown);
 * a unit too short to classify (detect() == None) reports CHIP_NEED_INPUT. */
enum { CHIP_UNKNOWN_FORMAT = 4 };

/* encoder::EncodeOp src/encoder/mod.rs:12-23, encoder::EncodeStatus src/encoder/mod.rs:27-38 */
enum { CHIP_OP_PROCESS = 0, CHIP_OP_FLUSH = 1, CHIP_OP_FINISH = 2 };
enum { CHIP_ENC_CONTINUE = 0, CHIP_ENC_NEED_OUTPUT = 1, CHIP_ENC_FINISHED = 2, CHIP_ENC_ERROR = 3 };

/* decoder::ZlibMode src/decoder/zlib_common.rs:4-15 and encoder ZlibMode
 * src/encoder/zlib_common.rs:28-37 use zlib's windowBits values; zstd gets its own tag. */
enum {
    CHIP_FMT_DEFLATE = -15,
    CHIP_FMT_ZLIB = 15,
    CHIP_FMT_GZIP = 31,
    CHIP_FMT_AUTO = 47, /* decoder only: zlib or gzip, src/decoder/zlib_common.rs:11-14 */
    CHIP_FMT_ZSTD = 100,
    /* chip_decode_batch only: route every unit by Detection::detect (src/decoder/mod.rs:28-114) to the
     * zlib/gzip or the zstd decoder -- the mixed gzip+zstd batch of BASELINE.json configs[4] */
    CHIP_FMT_DETECT = 0
};

/* decoder::Detection src/decoder/mod.rs:9-21; CHIP_DETECT_NONE is Rust's `None` (too few bytes) */
enum { CHIP_DETECT_NONE = -1, CHIP_DETECT_ZSTD = 0, CHIP_DETECT_GZIP = 1, CHIP_DETECT_ZLIB = 2, CHIP_DETECT_UNKNOWN = 3 };

/* library-level failures of the batch calls (not codec errors) */
enum { CHIP_OK = 0, CHIP_E_NO_DEVICE = -100, CHIP_E_INVALID = -101, CHIP_E_LAUNCH = -102, CHIP_E_NOMEM = -103 };

/* decoder::Decode src/decoder/mod.rs:150-157.  status is Ok(DecodeStatus) when err == 0 and
 * Err(DecodeError(err)) otherwise (err = zlib's negative code, or -(ZSTD_ErrorCode)). */
typedef struct {
    size_t input_remain;
    size_t output_remain;
    int32_t status;
    int32_t err;
} chip_decode_result;

/* encoder::Encode src/encoder/mod.rs:42-49 */
typedef struct {
    size_t input_remain;
    size_t output_remain;
    int32_t status;
} chip_encode_result;

/* ---- device / library ---------------------------------------------------------------------- */

/* Number of usable gfx950 devices (0 if none); never initialises more than the HIP runtime. */
int chip_device_count(void);
/* Select the device used by subsequent calls of this thread (hipSetDevice semantics). */
int chip_set_device(int device);
const char *chip_version(void);

/* src/mem.rs:27-76 routes every codec allocation through compu_malloc/compu_free.  Host-side
 * state of this backend goes through the same hooks when installed (signatures of
 * compu_malloc_with_state / compu_free_with_state, src/mem.rs:52-57,74-76); device memory comes
 * from hipMalloc and staging memory from hipHostMalloc. */
typedef void *(*chip_malloc_fn)(void *opaque, size_t size);
typedef void (*chip_free_fn)(void *opaque, void *ptr);
void chip_set_allocator(chip_malloc_fn malloc_fn, chip_free_fn free_fn, void *opaque);

/* Device and pinned-host buffers (north star: src/buffer.rs grows pinned-host + device types). */
void *chip_device_alloc(size_t size);
void chip_device_free(void *ptr);
void *chip_pinned_alloc(size_t size);
void chip_pinned_free(void *ptr);
int chip_memcpy_h2d(void *dst_dev, const void *src_host, size_t size, void *stream);
int chip_memcpy_d2h(void *dst_host, const void *src_dev, size_t size, void *stream);
int chip_stream_sync(void *stream);
/* The inflate kernel keeps a token scratch per (device, stream) it has been launched on: one 64 KiB slot per
 * resident wave, about 270 MB on an MI355X, allocated at the first launch and reused.  chip_trim() waits for the
 * current device and gives that memory back (the next launch allocates again).  No reference counterpart:
 * zlib-ng's inflate state is ~40 KiB of host memory per decoder (src/decoder/zlib_ng.rs:29-55). */
int chip_trim(void);

/* ---- streaming decoder: mirrors decoder::Interface, src/decoder/mod.rs:160-166 --------------- */

typedef struct chip_decoder chip_decoder;

typedef struct {
    int32_t window_log_max; /* ZstdOptions::window_log, src/decoder/zstd.rs:22-47; 0 = default */
    int32_t device;         /* HIP device ordinal, -1 = current */
} chip_decoder_opts;

/* Interface::zlib_ng(mode) src/decoder/zlib_ng.rs:61-90 / Interface::zstd(opts)
 * src/decoder/zstd.rs:81-94.  NULL on failure (the Rust side maps NULL to None). */
chip_decoder *chip_decoder_new(int format, const chip_decoder_opts *opts);
/* decode_fn: src/decoder/zlib_ng.rs:94-96 (+ macro src/decoder/mod.rs:459-486),
 * src/decoder/zstd.rs:98-136.  `in`/`out` are host pointers borrowed for the call only.
 * Limits: at most 256 MiB of compressed input may be BUFFERED (input behind the last deflate block boundary, or of
 * one zstd frame) and 4 GiB - 16 of decoded output held; a call that would cross them fails with err = -4
 * (Z_MEM_ERROR) instead of looping.  The calling thread's current HIP device is left as it was. */
chip_decode_result chip_decode(chip_decoder *d, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
/* reset_fn: src/decoder/zlib_ng.rs:99-108, src/decoder/zstd.rs:139-148.  Returns the instance
 * to keep using (compu replaces its pointer with the returned one, src/decoder/mod.rs:433-441). */
chip_decoder *chip_decoder_reset(chip_decoder *d);
/* drop_fn: src/decoder/zlib_ng.rs:111-115, src/decoder/zstd.rs:151-156 */
void chip_decoder_free(chip_decoder *d);
/* Memory a streaming decoder holds right now: pinned host bytes (buffered input) and device bytes (input copy, decoded
 * output not yet handed on + the 32 KiB window, checkpoint).  An inflate stream keeps O(window + piece) whatever its
 * length: input in front of the last block boundary and output that has been handed on are dropped between calls (the
 * reference's state is ~40 KiB per decoder, src/decoder/zlib_ng.rs:29-55).  A zstd stream keeps O(frame window + piece)
 * the same way (ZSTD_decompressStream's own buffers, src/decoder/zstd.rs:98-136): the block checkpoint carries the running
 * XXH64, output behind the window is dropped; a single-segment frame's window is its content size.  The inflate kernel's
 * token scratch (64 KiB per streaming decoder) is not included. */
void chip_decoder_footprint(const chip_decoder *d, size_t *pinned_bytes, size_t *device_bytes);
/* describe_error_fn: src/decoder/zlib_ng.rs:118-123 (zError), src/decoder/zstd.rs:159-164
 * (ZSTD_getErrorName).  Never NULL for code 0 (tests/decoder.rs:74-76). */
const char *chip_decoder_strerror(int format, int32_t code);

/* ---- batched decode: the hot path (additive API) -------------------------- */

/*
 * Decode n independent units in one launch, one wavefront per unit.  Every pointer is a DEVICE
 * pointer.  Unit i reads in_base[in_off[i] .. +in_len[i]) and writes out_base[out_off[i] ..
 * +out_cap[i]).  Results per unit:
 *   out_len[i]  bytes written
 *   in_used[i]  bytes of input consumed (trailing bytes after the stream are not counted)
 *   status[i]   CHIP_FINISHED / CHIP_NEED_INPUT (stream truncated) / CHIP_NEED_OUTPUT (out_cap too
 *               small) / CHIP_NEED_DICT, or a negative codec error with the meaning of DecodeError
 *               (zlib: -3 data error; zstd: -(ZSTD_ErrorCode), e.g. -20 corruption, -22 checksum)
 * For CHIP_FMT_ZSTD: on CHIP_NEED_OUTPUT out_len counts whole blocks only, and the unit's range up to
 * out_cap[i] may be used as scratch (regenerated literals are parked at its end while a block decodes).
 * A block is decoded in the unit's own range, so one that does not fit is CHIP_NEED_OUTPUT at the point
 * where the room ends -- also when damage lies further on in that block, which libzstd (decoding in a buffer of
 * its own) would report instead; with room for the frame the verdicts are the same.  A match offset beyond the
 * frame's Window_Size is -20 even where the bytes exist (RFC 8878 3.1.1.1.2): the verdict never depends on
 * how much history a streaming caller's decoder still holds.
 * in_base must be 4-byte aligned and its allocation padded to a multiple of 4 bytes.
 * `format` is one CHIP_FMT_* for the whole batch.  `stream` is a hipStream_t (NULL = default
 * stream); the call only enqueues work.  Returns CHIP_OK or a CHIP_E_* code.
 * Replaces, per unit, the loop  Interface::zlib_ng(mode) -> decode -> reset
 * (src/decoder/zlib_ng.rs:61-108, src/decoder/mod.rs:459-486).
 */
int chip_decode_batch(int format, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                      void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                      uint32_t *in_used, int32_t *status, void *stream);

/*
 * The same with options.  CHIP_F_COMPU_STATUS: status[i] is what compu's decode_fn would have returned for the unit, to the
 * letter, where the default names the cause instead --
 *   deflate / zlib / gzip (src/decoder/mod.rs:475-483): zlib's Z_OK with avail_in == 0 is NeedInput even when it was the
 *     output that filled (in_used[i] is then zlib's count: every bit of the token during which the room ran out, rounded up
 *     to a byte), and a unit without any input is NeedOutput (zlib's Z_BUF_ERROR);
 *   zstd (src/decoder/zstd.rs:121-133): compu compares output.pos with output.size before it looks at the return value -- a
 *     truncated frame whose bytes so far fill out_cap[i] exactly is NeedOutput; an error stays the error, also behind blocks
 *     that fill out_cap[i] exactly (ZSTD_decompressStream returns an error before it writes output.pos, so compu sees 0),
 *     except for an empty output range (0 == 0: NeedOutput).
 * Everything else is identical.  Unknown flag bits: CHIP_E_INVALID.
 */
enum { CHIP_F_COMPU_STATUS = 1 };
int chip_decode_batch_ex(int format, uint32_t flags, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                         void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                         uint32_t *in_used, int32_t *status, void *stream);

/*
 * The same for data that starts and ends in HOST memory (SURVEY.md sec. 8b "pinned-host variant", 8e): every
 * pointer is a host pointer (hipHostMalloc / chip_pinned_alloc memory lets the copies run asynchronously; pageable
 * memory works but serialises them).  The units are cut into slices (about `slice_bytes` of input + output each,
 * 0 = 256 MiB); slices alternate between two HIP streams, each running H2D -> kernel -> D2H for its slice, so the
 * copies of one slice overlap the kernel of the other.  Offsets are relative to in_base / out_base as in
 * chip_decode_batch.  A slice whose units lie back to back in index order (the usual layout; up to 15 bytes of
 * padding between units) moves straight between the caller's memory and the device; any other layout (gaps, reverse
 * order, overlapping ranges) is packed through pinned staging buffers.  Exactly out_len[i] bytes are written at
 * out_off[i]; nothing else in out_base is touched.  Returns when everything has arrived in host memory; the
 * calling thread's current device is left as it was.  Streams and buffers are kept per device between calls
 * (chip_trim() releases them).
 * No reference counterpart: compu's decode loop is the host-memory path (src/decoder/mod.rs:323-335).
 */
int chip_decode_batch_host(int format, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                           void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                           uint32_t *in_used, int32_t *status, int device, size_t slice_bytes);

/*
 * chip_decode_batch_host over several GPUs of the node (independent units, no exchange step, no
 * collective).  `devices[0 .. n_devices)` are HIP device ordinals (NULL / 0 = every visible device).  The units are
 * partitioned on the host into one contiguous index range per device, balanced by input + output bytes; a
 * CHIP_FMT_DETECT batch is first bucketed by Detection::detect (src/decoder/mod.rs:28-114) so that every launch is
 * homogeneous (gzip/zlib units to the inflate kernel, zstd frames to the zstd kernel; units that are neither are
 * answered on the host).  Each device is driven by a host thread of its own (hipSetDevice is per thread) with two
 * streams, as above; results land in the caller's per-unit arrays.  Returns the first failure of any device.
 */
int chip_decode_batch_multi(int format, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                            void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                            uint32_t *in_used, int32_t *status, const int *devices, int n_devices, size_t slice_bytes);

/* The host-side partition chip_decode_batch_multi uses: cuts[0 .. parts] with cuts[0] = 0 and
 * cuts[parts] = n; worker w owns units [cuts[w], cuts[w+1]), contiguous and balanced by in_len + out_cap bytes.
 * Pure host arithmetic (no device needed). */
int chip_partition_units(size_t n, const uint32_t *in_len, const uint32_t *out_cap, int parts, size_t *cuts);

/* Detection::detect src/decoder/mod.rs:28-114 on the first bytes of each unit; kind[i] gets a
 * CHIP_DETECT_* value.  Host form and batched device form. */
int chip_detect(const uint8_t *bytes, size_t len);
int chip_detect_batch(size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len, int32_t *kind,
                      void *stream);

/* ---- encoder: mirrors encoder::Interface, src/encoder/mod.rs:52-57 ---------------------------- */

typedef struct chip_encoder chip_encoder;

/* ZlibStrategy src/encoder/zlib_common.rs:5-24 (what zlib_ng.rs:69-75 hands to deflateInit2_) */
enum { CHIP_STRATEGY_DEFAULT = 0, CHIP_STRATEGY_FILTERED = 1, CHIP_STRATEGY_HUFFMAN_ONLY = 2, CHIP_STRATEGY_RLE = 3, CHIP_STRATEGY_FIXED = 4 };

/* ZlibOptions src/encoder/zlib_common.rs:47-103, every field (zlib-ng keeps them in its state and ignores the two
 * bytes compu replays on reset: src/encoder/zlib_ng.rs:84,95) */
typedef struct {
    int32_t mode;        /* CHIP_FMT_DEFLATE | CHIP_FMT_ZLIB | CHIP_FMT_GZIP (default Gzip, zlib_common.rs:33-37) */
    int32_t compression; /* 0..9, or -1 = zlib's default (6) as zlib_common.rs:96-103 allows: 0 stored blocks, 1 greedy
                          * match + one fixed-Huffman block, 2..9 the same match finder + dynamic-Huffman blocks,
                          * 4..9 with lazy choice (a match gives way to a longer one at the next position),
                          * 6..9 with two candidate positions per hash slot (the older one wins with a longer match) */
    int32_t device;      /* HIP device ordinal, -1 = current */
    int32_t strategy;    /* CHIP_STRATEGY_*: HuffmanOnly emits no matches, Rle only distance-1 matches, Fixed forces the
                          * fixed code at every level (zlib's Z_FIXED); Default and Filtered are the same here */
    int32_t mem_level;   /* 1..9 (zlib's memLevel; 0 = default 8): accepted for compatibility, the GPU state has one size */
} chip_encoder_opts;

/* Interface::zlib_ng(opts) src/encoder/zlib_ng.rs:50-87 */
chip_encoder *chip_encoder_new(const chip_encoder_opts *opts);
/* encode_fn src/encoder/zlib_ng.rs:90-92 (+ macro src/encoder/mod.rs:334-370) */
chip_encode_result chip_encode(chip_encoder *e, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int op);
/* reset_fn src/encoder/zlib_ng.rs:95-104 */
chip_encoder *chip_encoder_reset(chip_encoder *e);
/* drop_fn src/encoder/zlib_ng.rs:107-111 */
void chip_encoder_free(chip_encoder *e);

/* chip_encode_batch with a strategy (CHIP_STRATEGY_*); chip_encode_batch is strategy Default. */
int chip_encode_batch_ex(int format, int level, int strategy, size_t n, const void *in_base, const uint64_t *in_off,
                         const uint32_t *in_len, void *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                         uint32_t *out_len, int32_t *status, void *stream);

/* Batched level-1 encode of n independent units (device pointers, one wavefront per unit).
 * out_len[i] = compressed size; status[i] = CHIP_ENC_FINISHED or CHIP_ENC_NEED_OUTPUT.  Each unit becomes
 * one complete stream of `format` (wrapper, one fixed-Huffman or stored deflate body, trailer).  The
 * range out_off[i] .. +out_cap[i] may be used as scratch beyond out_len[i]. */
int chip_encode_batch(int format, int level, size_t n, const void *in_base, const uint64_t *in_off,
                      const uint32_t *in_len, void *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                      uint32_t *out_len, int32_t *status, void *stream);
/* Worst-case compressed size for in_len input bytes in `format` (sizing out_cap). */
size_t chip_encode_bound(int format, size_t in_len);

/* chip_encode_batch for data in (pinned) host memory: same two-stream slicing and layout rules as
 * chip_decode_batch_host (exactly out_len[i] bytes are written at out_off[i]). */
int chip_encode_batch_host(int format, int level, size_t n, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                           void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, int32_t *status,
                           int device, size_t slice_bytes);

#ifdef __cplusplus
}
#endif
#endif
