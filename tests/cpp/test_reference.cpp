// The reference's integration tests replayed against the `hip` Interface variant through the C++ mirror
// (compu_amd/host/compu.hpp) and therefore through the C ABI:
//   tests/decoder.rs:21-77  test_case            (zlib-ng gzip :141-150, zstd :119-128)
//   tests/encoder.rs:10-78  test_case            (zlib-ng gzip :216-225, zlib :249-258, deflate :282-291)
//   tests/encoder.rs:115-173 test_case_empty_final (:345-354, :378-387, :411-420)
// Usage: test_reference <dir with 10x10y, alice29.txt and their .compressed.{gz,zstd}>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../compu_amd/host/compu.hpp"

using namespace compu;
using decoder::DecodeError;
using decoder::DecodeStatus;
using decoder::Decoder;
using decoder::Detection;
using encoder::EncodeOp;
using encoder::Encoder;
using encoder::EncodeStatus;

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            std::fprintf(stderr, "%s:%d: assertion failed: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

static std::vector<uint8_t> read_file(const std::string &path)
{
    std::vector<uint8_t> v;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path.c_str());
        std::exit(2);
    }
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    std::fclose(f);
    return v;
}

static const size_t DATA_LEN = 2;  // DATA is a two-element array in the reference, so DATA.len() / 2 == 1

// tests/decoder.rs:21-77
static void decoder_test_case(Decoder &decoder, const std::vector<uint8_t> &data, const std::vector<uint8_t> &compressed)
{
    // Full
    std::vector<uint8_t> output(data.size(), 0);
    auto result = decoder.decode(compressed.data(), compressed.size(), output.data(), output.size());
    CHECK(result.is(DecodeStatus::Finished));
    CHECK(result.input_remain == 0);
    CHECK(result.output_remain == 0);
    CHECK(data == output);
    decoder.reset();

    // Partial buffer
    const size_t half = DATA_LEN / 2;
    result = decoder.decode(compressed.data(), compressed.size(), output.data(), half);
    CHECK(result.is(DecodeStatus::NeedOutput));
    CHECK(result.output_remain == 0);
    const uint8_t *remaining = compressed.data() + (compressed.size() - result.input_remain);
    result = decoder.decode(remaining, result.input_remain, output.data() + half, output.size() - half);
    CHECK(result.is(DecodeStatus::Finished));
    CHECK(data == output);
    decoder.reset();

    // Buffered decoder
    Buffer<4096> buffer;
    const uint8_t *buffer_input = compressed.data();
    size_t buffer_input_len = compressed.size();
    output.clear();
    for (;;) {
        DecodeError err;
        auto r = buffer.decode(decoder, buffer_input, buffer_input_len, &err);
        CHECK(r.first);
        buffer_input += r.second.first;
        buffer_input_len -= r.second.first;
        output.insert(output.end(), buffer.data(), buffer.data() + buffer.len());
        buffer.consume();
        if (r.second.second == DecodeStatus::Finished) break;
    }
    CHECK(data == output);
    decoder.reset();

    // Buffered decoder over page-locked memory: the hip backend's PinnedBuffer with Buffer<N>'s cursor API
    {
        PinnedBuffer pinned(4096);
        CHECK(pinned.valid() && pinned.capacity() == 4096 && pinned.len() == 0);
        buffer_input = compressed.data();
        buffer_input_len = compressed.size();
        output.clear();
        for (;;) {
            DecodeError err;
            auto r = pinned.decode(decoder, buffer_input, buffer_input_len, &err);
            CHECK(r.first);
            buffer_input += r.second.first;
            buffer_input_len -= r.second.first;
            output.insert(output.end(), pinned.data(), pinned.data() + pinned.len());
            pinned.consume();
            if (r.second.second == DecodeStatus::Finished) break;
        }
        CHECK(data == output);
        decoder.reset();
    }

    // Full vec
    output.clear();
    output.shrink_to_fit();
    result = decoder.decode_vec_full(compressed.data(), compressed.size(), output);
    CHECK(result.is(DecodeStatus::Finished));
    CHECK(result.input_remain == 0);
    CHECK(data == output);
    decoder.reset();

    CHECK(decoder.describe_error(DecodeError::no_error()) != nullptr);
}

// The hip backend's DeviceBuffer: a batch is decoded from device memory into device memory behind the buffer's cursor.
static void device_buffer_case(const std::vector<uint8_t> &compressed, const std::vector<uint8_t> &data, int format)
{
    const size_t n = 3;  // the same stream three times, as three units
    const size_t in_stride = (compressed.size() + 3) & ~(size_t)3, out_stride = (data.size() + 15) & ~(size_t)15;
    DeviceBuffer in(n * in_stride), out(64 + n * out_stride), arrays(n * 40 + 64);
    CHECK(in.valid() && out.valid() && arrays.valid());
    std::vector<uint8_t> padded(in_stride, 0);
    memcpy(padded.data(), compressed.data(), compressed.size());
    for (size_t i = 0; i < n; i++) CHECK(in.upload(padded.data(), padded.size()));
    CHECK(in.len() == n * in_stride && in.spare_capacity_len() == 0);
    // per-unit arrays: in_off u64[n], out_off u64[n], in_len u32[n], out_cap u32[n], then results out_len, in_used, status
    std::vector<uint8_t> host(n * 40, 0);
    uint64_t *in_off = (uint64_t *)host.data(), *out_off = in_off + n;
    uint32_t *in_len = (uint32_t *)(out_off + n), *out_cap = in_len + n;
    for (size_t i = 0; i < n; i++) {
        in_off[i] = i * in_stride;
        out_off[i] = i * out_stride;
        in_len[i] = (uint32_t)compressed.size();
        out_cap[i] = (uint32_t)data.size();
    }
    CHECK(arrays.upload(host.data(), host.size()));
    const uint8_t *d = arrays.data();
    const uint64_t *d_in_off = (const uint64_t *)d, *d_out_off = d_in_off + n;
    const uint32_t *d_in_len = (const uint32_t *)(d_out_off + n), *d_out_cap = d_in_len + n;
    uint32_t *d_res = (uint32_t *)(d_out_cap + n);
    const uint8_t marker[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
    CHECK(out.upload(marker, sizeof marker));  // the batch lands BEHIND what the buffer already holds
    CHECK(out.decode_batch(format, n, in, d_in_off, d_in_len, d_out_off, d_out_cap, d_res, d_res + n, (int32_t *)(d_res + 2 * n), n * out_stride) == CHIP_OK);
    CHECK(chip_stream_sync(nullptr) == CHIP_OK);
    CHECK(out.len() == 16 + n * out_stride);
    std::vector<uint8_t> back(out.len());
    CHECK(out.download(back.data(), 0, back.size()));
    CHECK(memcmp(back.data(), marker, 16) == 0);
    for (size_t i = 0; i < n; i++) CHECK(memcmp(back.data() + 16 + i * out_stride, data.data(), data.size()) == 0);
    std::vector<uint8_t> res(n * 12);
    CHECK(chip_memcpy_d2h(res.data(), d_res, res.size(), nullptr) == CHIP_OK && chip_stream_sync(nullptr) == CHIP_OK);
    const uint32_t *r = (const uint32_t *)res.data();
    for (size_t i = 0; i < n; i++) CHECK(r[i] == data.size() && r[n + i] == compressed.size() && (int32_t)r[2 * n + i] == CHIP_FINISHED);
    out.consume();
    CHECK(out.len() == 0 && out.spare_capacity_len() == out.capacity());
}

// tests/encoder.rs:10-78
static void encoder_test_case(Encoder &encoder, Decoder &decoder, const std::vector<uint8_t> &data, Detection expected_detection)
{
    std::vector<uint8_t> compressed(data.size(), 0), compressed_full, decompressed(data.size(), 0), decompressed_full;
    auto result = encoder.encode(data.data(), data.size(), compressed.data(), compressed.size(), EncodeOp::Finish);
    CHECK(result.input_remain == 0);
    if (result.status == EncodeStatus::NeedOutput) {
        const size_t len = compressed.size();
        compressed.resize(len + 100);
        result = encoder.encode(nullptr, 0, compressed.data() + len, 100, EncodeOp::Finish);
        CHECK(result.status == EncodeStatus::Finished);
        compressed.resize(len + 100 - result.output_remain);
    } else {
        compressed.resize(compressed.size() - result.output_remain);
    }
    auto det = decoder::detect(compressed.data(), compressed.size());
    CHECK(det.has_value() && *det == expected_detection);
    auto dres = decoder.decode(compressed.data(), compressed.size(), decompressed.data(), decompressed.size());
    CHECK(dres.is(DecodeStatus::Finished));
    CHECK(data == decompressed);

    // Buffered encoder
    encoder.reset();
    Buffer<4096> buffer;
    const uint8_t *buffer_input = data.data();
    size_t buffer_input_len = data.size();
    for (;;) {
        auto r = buffer.encode(encoder, buffer_input, buffer_input_len, EncodeOp::Finish);
        buffer_input += r.first;
        buffer_input_len -= r.first;
        compressed_full.insert(compressed_full.end(), buffer.data(), buffer.data() + buffer.len());
        buffer.consume();
        CHECK(r.second != EncodeStatus::Error);
        if (r.second == EncodeStatus::Finished) break;
    }
    CHECK(compressed == compressed_full);
    compressed_full.clear();

    // Full vec encoding
    encoder.reset();
    result = encoder.encode_vec_full(data.data(), data.size(), compressed_full, EncodeOp::Finish);
    CHECK(result.status == EncodeStatus::Finished);
    CHECK(result.input_remain == 0);
    CHECK(compressed == compressed_full);

    decoder.reset();
    dres = decoder.decode_vec_full(compressed_full.data(), compressed_full.size(), decompressed_full);
    CHECK(dres.is(DecodeStatus::Finished));
    CHECK(data == decompressed_full);
    encoder.reset();
    decoder.reset();
}

// tests/encoder.rs:115-173
static void encoder_test_case_empty_final(Encoder &encoder, Decoder &decoder, const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> compressed;
    compressed.reserve(data.size());
    auto result = encoder.encode_vec(data.data(), data.size(), compressed, EncodeOp::Process);
    CHECK(result.status != EncodeStatus::Error);
    result = encoder.encode_vec(data.data() + (data.size() - result.input_remain), result.input_remain, compressed, EncodeOp::Flush);
    CHECK(result.input_remain == 0);
    CHECK(result.status == EncodeStatus::Continue);
    compressed.reserve(compressed.size() + 100);
    result = encoder.encode_vec(nullptr, 0, compressed, EncodeOp::Finish);
    if (result.status == EncodeStatus::NeedOutput)  // room the reference gets from zlib-ng's ratio; see tests/test_encoder_gpu.py
        result = encoder.encode_vec_full(nullptr, 0, compressed, EncodeOp::Finish);
    CHECK(result.status == EncodeStatus::Finished);

    std::vector<uint8_t> decompressed(data.size() + 100, 0);
    size_t got = 0;
    const size_t step = compressed.size() / 4 ? compressed.size() / 4 : 1;
    for (size_t off = 0; off < compressed.size(); off += step) {
        const size_t n = compressed.size() - off < step ? compressed.size() - off : step;
        auto r = decoder.decode(compressed.data() + off, n, decompressed.data() + got, decompressed.size() - got);
        CHECK(r.input_remain == 0);
        CHECK(r.output_remain > 0);
        got = decompressed.size() - r.output_remain;
        CHECK(r.ok);
        if (r.status == DecodeStatus::Finished) break;
        CHECK(r.status == DecodeStatus::NeedInput);
    }
    decompressed.resize(got);
    CHECK(data == decompressed);
    encoder.reset();
    decoder.reset();
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "tests/golden";
    const char *names[2] = {"10x10y", "alice29.txt"};
    std::vector<uint8_t> DATA[2], DATA_GZIP[2], DATA_ZSTD[2];
    for (int i = 0; i < 2; i++) {
        DATA[i] = read_file(dir + "/" + names[i]);
        DATA_GZIP[i] = read_file(dir + "/" + names[i] + ".compressed.gz");
        DATA_ZSTD[i] = read_file(dir + "/" + names[i] + ".compressed.zstd");
    }
    if (chip_device_count() < 1) {
        std::fprintf(stderr, "no HIP device: the hip backend has no CPU path\n");
        return 3;
    }
    {  // should_decode_zlib_ng_gzip, tests/decoder.rs:141-150
        auto decoder = decoder::Interface::zlib_hip(decoder::ZlibMode::Gzip);
        CHECK(decoder.has_value());
        for (int i = 0; i < 2; i++) decoder_test_case(*decoder, DATA[i], DATA_GZIP[i]);
        std::puts("should_decode_zlib_hip_gzip ... ok");
    }
    {  // should_decode_zstd, tests/decoder.rs:119-128
        auto decoder = decoder::Interface::zstd_hip();
        CHECK(decoder.has_value());
        for (int i = 0; i < 2; i++) decoder_test_case(*decoder, DATA[i], DATA_ZSTD[i]);
        std::puts("should_decode_zstd_hip ... ok");
    }
    struct {
        decoder::ZlibMode mode;
        Detection det;
        const char *name;
    } modes[3] = {{decoder::ZlibMode::Gzip, Detection::Gzip, "gzip"}, {decoder::ZlibMode::Zlib, Detection::Zlib, "zlib"},
                  {decoder::ZlibMode::Deflate, Detection::Unknown, "deflate"}};
    for (auto &m : modes) {  // tests/encoder.rs:216-225, 249-258, 282-291 and the empty-final variants
        auto enc = encoder::Interface::zlib_hip(encoder::ZlibOptions().mode(m.mode).compression(1));
        auto dec = decoder::Interface::zlib_hip(m.mode);
        CHECK(enc.has_value() && dec.has_value());
        for (int i = 0; i < 2; i++) encoder_test_case(*enc, *dec, DATA[i], m.det);
        for (int i = 0; i < 2; i++) encoder_test_case_empty_final(*enc, *dec, DATA[i]);
        std::printf("should_encode_and_decode_zlib_hip_%s (+ empty_final) ... ok\n", m.name);
    }
    {  // the hip backend's device buffer type (north star: src/buffer.rs grows pinned-host + device buffer types)
        for (int i = 0; i < 2; i++) {
            device_buffer_case(DATA_GZIP[i], DATA[i], CHIP_FMT_GZIP);
            device_buffer_case(DATA_ZSTD[i], DATA[i], CHIP_FMT_ZSTD);
        }
        std::puts("device_buffer_batch_decode ... ok");
    }
    std::puts("test result: ok");
    return 0;
}
