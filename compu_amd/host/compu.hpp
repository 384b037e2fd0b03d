// compu.hpp -- C++ mirror of compu's Decoder / Encoder surface over the C ABI of libcompu_hip.so.
//
// The reference is a Rust crate; this image has no Rust toolchain, so the host side above the C ABI is
// written in C++ with the reference's names, argument meaning and error behaviour:
//   compu::decoder::{Interface, Decoder, Decode, DecodeStatus, DecodeError, Detection, ZlibMode, ZstdOptions}
//       <- src/decoder/mod.rs, src/decoder/zlib_common.rs, src/decoder/zstd.rs
//   compu::encoder::{Interface, Encoder, Encode, EncodeOp, EncodeStatus, ZlibOptions}
//       <- src/encoder/mod.rs, src/encoder/zlib_common.rs
//   compu::Buffer<N>, compu::PinnedBuffer, compu::DeviceBuffer          <- src/buffer.rs (+ the north star's pinned-host / device types)
// so that tests/cpp/test_reference.cpp reads like tests/decoder.rs / tests/encoder.rs.  The Rust glue a
// maintainer would add to the crate itself is in INTEGRATION.md.
#pragma once

#include <cstddef>
#include <cstdint>
#include <optional>
#include <utility>
#include <vector>

#include "../../include/compu_hip.h"

namespace compu {

template <size_t N>
class Buffer;

namespace decoder {

// src/decoder/mod.rs:9-21, 28-114
enum class Detection { Zstd = CHIP_DETECT_ZSTD, Gzip = CHIP_DETECT_GZIP, Zlib = CHIP_DETECT_ZLIB, Unknown = CHIP_DETECT_UNKNOWN };
inline std::optional<Detection> detect(const uint8_t *bytes, size_t len)
{
    int k = chip_detect(bytes, len);
    if (k < 0) return std::nullopt;
    return static_cast<Detection>(k);
}

// src/decoder/mod.rs:117-135
struct DecodeError {
    int32_t code = 0;
    static DecodeError no_error() { return DecodeError{0}; }
    int32_t as_raw() const { return code; }
    bool operator==(const DecodeError &o) const { return code == o.code; }
};

// src/decoder/mod.rs:139-146
enum class DecodeStatus { NeedInput = CHIP_NEED_INPUT, NeedOutput = CHIP_NEED_OUTPUT, Finished = CHIP_FINISHED };

// src/decoder/mod.rs:150-157: status is Result<DecodeStatus, DecodeError>
struct Decode {
    size_t input_remain;
    size_t output_remain;
    bool ok;
    DecodeStatus status;  // valid when ok
    DecodeError error;    // valid when !ok
    bool is(DecodeStatus s) const { return ok && status == s; }
};

// src/decoder/zlib_common.rs:4-29
enum class ZlibMode : int { Deflate = CHIP_FMT_DEFLATE, Zlib = CHIP_FMT_ZLIB, Gzip = CHIP_FMT_GZIP, Auto = CHIP_FMT_AUTO };

// src/decoder/zstd.rs:22-74
struct ZstdOptions {
    int32_t window_log_ = 0;
    ZstdOptions window_log(int32_t v) const
    {
        ZstdOptions o = *this;
        o.window_log_ = v;
        return o;
    }
};

// src/decoder/mod.rs:269-455
class Decoder {
public:
    Decoder(chip_decoder *h, int fmt) : h_(h), fmt_(fmt) {}
    Decoder(Decoder &&o) noexcept : h_(o.h_), fmt_(o.fmt_) { o.h_ = nullptr; }
    Decoder(const Decoder &) = delete;
    Decoder &operator=(const Decoder &) = delete;
    ~Decoder() { chip_decoder_free(h_); }  // Drop, mod.rs:450-455

    // raw_decode, mod.rs:290-292
    Decode raw_decode(const uint8_t *input, size_t input_len, uint8_t *output, size_t output_len)
    {
        chip_decode_result r = chip_decode(h_, input, input_len, output, output_len);
        Decode d{r.input_remain, r.output_remain, r.err == 0, DecodeStatus::NeedInput, DecodeError{r.err}};
        if (d.ok) d.status = static_cast<DecodeStatus>(r.status);
        return d;
    }
    // decode, mod.rs:309-315
    Decode decode(const uint8_t *input, size_t input_len, uint8_t *output, size_t output_len)
    {
        static uint8_t dummy_in = 0, dummy_out = 0;  // slices are never null in Rust
        return raw_decode(input_len ? input : &dummy_in, input_len, output_len ? output : &dummy_out, output_len);
    }
    // decode_vec, mod.rs:323-335: writes into the spare capacity, advances len only on Ok
    Decode decode_vec(const uint8_t *input, size_t input_len, std::vector<uint8_t> &output)
    {
        const size_t len = output.size(), spare = output.capacity() - len;
        output.resize(output.capacity());
        Decode r = decode(input, input_len, output.data() + len, spare);
        output.resize(r.ok ? len + spare - r.output_remain : len);
        return r;
    }
    // decode_vec_full, mod.rs:360-385
    Decode decode_vec_full(const uint8_t *input, size_t input_len, std::vector<uint8_t> &output)
    {
        constexpr size_t RESERVE_DEFAULT = 1024;
        size_t reserve_size;
        if (input_len < RESERVE_DEFAULT) {
            output.reserve(output.size() + input_len);
            reserve_size = input_len / 3;
        } else if (input_len < RESERVE_DEFAULT * 16) {
            output.reserve(output.size() + input_len + input_len / 3);
            reserve_size = RESERVE_DEFAULT;
        } else {
            output.reserve(output.size() + input_len * 2);
            reserve_size = RESERVE_DEFAULT * 8;
        }
        for (;;) {
            Decode r = decode_vec(input, input_len, output);
            if (r.is(DecodeStatus::NeedOutput)) {
                input += input_len - r.input_remain;
                input_len = r.input_remain;
                output.reserve(output.size() + (reserve_size ? reserve_size : 1));
                continue;
            }
            return r;
        }
    }
    // reset, mod.rs:433-441: the returned instance replaces the held one
    bool reset()
    {
        chip_decoder *n = chip_decoder_reset(h_);
        if (!n) return false;
        h_ = n;
        return true;
    }
    // describe_error, mod.rs:445-447
    const char *describe_error(DecodeError e) const { return chip_decoder_strerror(fmt_, e.as_raw()); }

private:
    chip_decoder *h_;
    int fmt_;
};

// decoder::Interface constructors of the `hip` variant
struct Interface {
    // Interface::zlib_ng(mode), src/decoder/zlib_ng.rs:61-90
    static std::optional<Decoder> zlib_hip(ZlibMode mode = ZlibMode::Auto, int device = -1)
    {
        chip_decoder_opts o{0, device};
        chip_decoder *h = chip_decoder_new(static_cast<int>(mode), &o);
        if (!h) return std::nullopt;
        return Decoder(h, static_cast<int>(mode));
    }
    // Interface::zstd(opts), src/decoder/zstd.rs:81-94
    static std::optional<Decoder> zstd_hip(ZstdOptions opts = {}, int device = -1)
    {
        chip_decoder_opts o{opts.window_log_, device};
        chip_decoder *h = chip_decoder_new(CHIP_FMT_ZSTD, &o);
        if (!h) return std::nullopt;
        return Decoder(h, CHIP_FMT_ZSTD);
    }
};

}  // namespace decoder

namespace encoder {

enum class EncodeOp { Process = CHIP_OP_PROCESS, Flush = CHIP_OP_FLUSH, Finish = CHIP_OP_FINISH };  // mod.rs:12-23
enum class EncodeStatus { Continue = CHIP_ENC_CONTINUE, NeedOutput = CHIP_ENC_NEED_OUTPUT, Finished = CHIP_ENC_FINISHED, Error = CHIP_ENC_ERROR };  // mod.rs:27-38

// src/encoder/mod.rs:42-49
struct Encode {
    size_t input_remain;
    size_t output_remain;
    EncodeStatus status;
};

using ZlibMode = decoder::ZlibMode;  // src/encoder/zlib_common.rs:28-37 (Deflate | Zlib | Gzip)

// src/encoder/zlib_common.rs:5-24
enum class ZlibStrategy { Default = 0, Filtered = 1, HuffmanOnly = 2, Rle = 3, Fixed = 4 };

// src/encoder/zlib_common.rs:47-103 (defaults: Gzip, Default strategy, mem_level 8, compression 9, :59-66)
struct ZlibOptions {
    ZlibMode mode_ = ZlibMode::Gzip;
    ZlibStrategy strategy_ = ZlibStrategy::Default;
    int mem_level_ = 8;
    int compression_ = 9;
    ZlibOptions strategy(ZlibStrategy s) const
    {
        ZlibOptions o = *this;
        o.strategy_ = s;
        return o;
    }
    ZlibOptions mem_level(int level) const
    {
        ZlibOptions o = *this;
        o.mem_level_ = level;
        return o;
    }
    ZlibOptions mode(ZlibMode m) const
    {
        ZlibOptions o = *this;
        o.mode_ = m;
        return o;
    }
    ZlibOptions compression(int level) const
    {
        ZlibOptions o = *this;
        o.compression_ = level;
        return o;
    }
};

// src/encoder/mod.rs:148-330
class Encoder {
public:
    explicit Encoder(chip_encoder *h) : h_(h) {}
    Encoder(Encoder &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Encoder(const Encoder &) = delete;
    Encoder &operator=(const Encoder &) = delete;
    ~Encoder() { chip_encoder_free(h_); }

    // raw_encode / encode, mod.rs:171-199
    Encode encode(const uint8_t *input, size_t input_len, uint8_t *output, size_t output_len, EncodeOp op)
    {
        static uint8_t dummy_in = 0, dummy_out = 0;
        chip_encode_result r = chip_encode(h_, input_len ? input : &dummy_in, input_len, output_len ? output : &dummy_out, output_len, static_cast<int>(op));
        return Encode{r.input_remain, r.output_remain, static_cast<EncodeStatus>(r.status)};
    }
    // encode_vec, mod.rs:203-213: the length is set even on Error
    Encode encode_vec(const uint8_t *input, size_t input_len, std::vector<uint8_t> &output, EncodeOp op)
    {
        const size_t len = output.size(), spare = output.capacity() - len;
        output.resize(output.capacity());
        Encode r = encode(input, input_len, output.data() + len, spare, op);
        output.resize(len + spare - r.output_remain);
        return r;
    }
    // encode_vec_full, mod.rs:239-267
    Encode encode_vec_full(const uint8_t *input, size_t input_len, std::vector<uint8_t> &output, EncodeOp op)
    {
        constexpr size_t RESERVE_DEFAULT = 1024;
        size_t reserve_size;
        if (input_len < RESERVE_DEFAULT) {
            output.reserve(output.size() + input_len);
            reserve_size = input_len / 3;
        } else if (input_len < RESERVE_DEFAULT * 16) {
            output.reserve(output.size() + input_len / 2);
            reserve_size = RESERVE_DEFAULT;
        } else {
            output.reserve(output.size() + input_len / 3);
            reserve_size = RESERVE_DEFAULT * 8;
        }
        for (;;) {
            Encode r = encode_vec(input, input_len, output, op);
            if (r.status == EncodeStatus::NeedOutput) {
                input += input_len - r.input_remain;
                input_len = r.input_remain;
                output.reserve(output.size() + (reserve_size ? reserve_size : 1));
                continue;
            }
            if (r.status == EncodeStatus::Continue && op == EncodeOp::Finish) {
                input += input_len - r.input_remain;
                input_len = r.input_remain;
                continue;
            }
            return r;
        }
    }
    // reset, mod.rs:314-321
    bool reset()
    {
        chip_encoder *n = chip_encoder_reset(h_);
        if (!n) return false;
        h_ = n;
        return true;
    }

private:
    chip_encoder *h_;
};

struct Interface {
    // Interface::zlib_ng(opts), src/encoder/zlib_ng.rs:50-87
    static std::optional<Encoder> zlib_hip(ZlibOptions opts = {}, int device = -1)
    {
        chip_encoder_opts o{static_cast<int32_t>(opts.mode_), opts.compression_, device, static_cast<int32_t>(opts.strategy_), opts.mem_level_};
        chip_encoder *h = chip_encoder_new(&o);
        if (!h) return std::nullopt;
        return Encoder(h);
    }
};

}  // namespace encoder

// src/buffer.rs:1-49 with Buffer::decode (src/decoder/mod.rs:507-531) and Buffer::encode
// (src/encoder/mod.rs:395-412)
template <size_t N>
class Buffer {
public:
    const uint8_t *data() const { return buf_; }
    size_t len() const { return cursor_; }
    void consume() { cursor_ = 0; }

    // Ok((consumed, status)) or Err(error); on error the buffer does not advance
    std::pair<bool, std::pair<size_t, decoder::DecodeStatus>> decode(decoder::Decoder &dec, const uint8_t *input, size_t input_len,
                                                                      decoder::DecodeError *err = nullptr)
    {
        const size_t spare = N - cursor_;
        decoder::Decode r = dec.decode(input, input_len, buf_ + cursor_, spare);
        if (!r.ok) {
            if (err) *err = r.error;
            return {false, {0, decoder::DecodeStatus::NeedInput}};
        }
        cursor_ += spare - r.output_remain;
        return {true, {input_len - r.input_remain, r.status}};
    }
    std::pair<size_t, encoder::EncodeStatus> encode(encoder::Encoder &enc, const uint8_t *input, size_t input_len, encoder::EncodeOp op)
    {
        const size_t spare = N - cursor_;
        encoder::Encode r = enc.encode(input, input_len, buf_ + cursor_, spare, op);
        cursor_ += spare - r.output_remain;
        return {input_len - r.input_remain, r.status};
    }

private:
    uint8_t buf_[N];
    size_t cursor_ = 0;
};

// The two buffer types the hip backend adds to src/buffer.rs (north star: "src/buffer.rs grows pinned-host + device
// buffer types"), with Buffer<N>'s cursor API (data / len / consume / spare capacity; src/buffer.rs:9-49).
//
// PinnedBuffer: page-locked host memory (hipHostMalloc through chip_pinned_alloc).  Used exactly like Buffer<N> with the
// streaming Decoder / Encoder; the copies between it and the GPU are real DMA transfers instead of staged ones.
class PinnedBuffer {
public:
    explicit PinnedBuffer(size_t capacity) : buf_((uint8_t *)chip_pinned_alloc(capacity)), cap_(buf_ ? capacity : 0) {}
    ~PinnedBuffer() { chip_pinned_free(buf_); }
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    bool valid() const { return buf_ != nullptr; }
    const uint8_t *data() const { return buf_; }
    size_t len() const { return cursor_; }
    size_t capacity() const { return cap_; }
    void consume() { cursor_ = 0; }
    uint8_t *spare_capacity_mut() { return buf_ + cursor_; }
    size_t spare_capacity_len() const { return cap_ - cursor_; }
    void advance(size_t n) { cursor_ += n; }  // after writing n bytes into the spare capacity

    std::pair<bool, std::pair<size_t, decoder::DecodeStatus>> decode(decoder::Decoder &dec, const uint8_t *input, size_t input_len,
                                                                      decoder::DecodeError *err = nullptr)
    {
        const size_t spare = cap_ - cursor_;
        decoder::Decode r = dec.decode(input, input_len, buf_ + cursor_, spare);
        if (!r.ok) {
            if (err) *err = r.error;
            return {false, {0, decoder::DecodeStatus::NeedInput}};
        }
        cursor_ += spare - r.output_remain;
        return {true, {input_len - r.input_remain, r.status}};
    }
    std::pair<size_t, encoder::EncodeStatus> encode(encoder::Encoder &enc, const uint8_t *input, size_t input_len, encoder::EncodeOp op)
    {
        const size_t spare = cap_ - cursor_;
        encoder::Encode r = enc.encode(input, input_len, buf_ + cursor_, spare, op);
        cursor_ += spare - r.output_remain;
        return {input_len - r.input_remain, r.status};
    }

private:
    uint8_t *buf_;
    size_t cap_;
    size_t cursor_ = 0;
};

// DeviceBuffer: memory of the current GPU (hipMalloc through chip_device_alloc; src/mem.rs routes device allocations
// there).  The batched entry points read and write device memory, so a DeviceBuffer is what a batch decodes out of and
// into without the data ever touching the host: upload() appends host bytes, decode_batch() appends the decoded units
// behind the cursor, download() reads back.
class DeviceBuffer {
public:
    explicit DeviceBuffer(size_t capacity) : buf_((uint8_t *)chip_device_alloc(capacity + 16)), cap_(buf_ ? capacity : 0) {}
    ~DeviceBuffer() { chip_device_free(buf_); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    bool valid() const { return buf_ != nullptr; }
    const uint8_t *data() const { return buf_; }  // DEVICE pointer to the written part
    size_t len() const { return cursor_; }
    size_t capacity() const { return cap_; }
    void consume() { cursor_ = 0; }
    uint8_t *spare_capacity_mut() { return buf_ + cursor_; }  // DEVICE pointer
    size_t spare_capacity_len() const { return cap_ - cursor_; }
    void advance(size_t n) { cursor_ += n; }

    // append host bytes (synchronous); false when they do not fit
    bool upload(const uint8_t *src, size_t n)
    {
        if (n > cap_ - cursor_) return false;
        if (n && (chip_memcpy_h2d(buf_ + cursor_, src, n, nullptr) != CHIP_OK || chip_stream_sync(nullptr) != CHIP_OK)) return false;
        cursor_ += n;
        return true;
    }
    // copy `n` written bytes from offset `from` to the host (synchronous)
    bool download(uint8_t *dst, size_t from, size_t n) const
    {
        if (from + n > cursor_) return false;
        return n == 0 || (chip_memcpy_d2h(dst, buf_ + from, n, nullptr) == CHIP_OK && chip_stream_sync(nullptr) == CHIP_OK);
    }
    // chip_decode_batch[_ex] with this buffer's spare capacity as the output: unit i lands at spare + out_off[i] (device
    // arrays, as in chip_decode_batch); `span` bytes behind the cursor become part of the data.  Only enqueues.
    // flags: CHIP_F_COMPU_STATUS reports every unit's status exactly as compu's decode_fn would (include/compu_hip.h).
    int decode_batch(int format, size_t n, const DeviceBuffer &in, const uint64_t *in_off, const uint32_t *in_len, const uint64_t *out_off,
                     const uint32_t *out_cap, uint32_t *out_len, uint32_t *in_used, int32_t *status, size_t span, void *stream = nullptr,
                     uint32_t flags = 0)
    {
        if (span > cap_ - cursor_) return CHIP_E_INVALID;
        const int rc = chip_decode_batch_ex(format, flags, n, in.data(), in_off, in_len, buf_ + cursor_, out_off, out_cap, out_len, in_used, status, stream);
        if (rc == CHIP_OK) cursor_ += span;
        return rc;
    }

private:
    uint8_t *buf_;
    size_t cap_;
    size_t cursor_ = 0;
};

}  // namespace compu
