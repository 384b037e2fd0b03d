// Batched form of Detection::detect (src/decoder/mod.rs:28-114): classify each unit by its first
// 2-4 bytes.  One lane per unit; the router in front of a mixed gzip+zstd batch.
#include "chip_internal.h"

namespace chip {

__global__ void detect_kernel(uint32_t n, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len, int32_t *kind)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *b = in_base + in_off[i];
    uint32_t len = in_len[i];
    int32_t k = CHIP_DETECT_NONE;
    if (len >= 2) {
        uint32_t b0 = b[0], b1 = b[1];
        if (b0 == 0x1f && b1 == 0x8b) k = CHIP_DETECT_GZIP;
        else {
            bool zl = false;
            if (((b0 << 8) | b1) % 31 == 0 && (b0 & 0x8f) == 0x08 && b0 != 0x68) {
                // FLG values of the table at src/decoder/mod.rs:44-55, one packed word per CINFO;
                // the 0x68 row never matches in the reference (mod.rs:80-82) and is kept that way
                const uint32_t rows[8] = {0x1d5b99d7u, 0x195795d3u, 0x155391cfu, 0x114f8dcbu,
                                          0x0d4b89c7u, 0x094785c3u, 0x054381deu, 0x015e9cdau};
                uint32_t r = rows[b0 >> 4];
                zl = b1 == (r >> 24) || b1 == ((r >> 16) & 0xff) || b1 == ((r >> 8) & 0xff) || b1 == (r & 0xff);
            }
            if (zl) k = CHIP_DETECT_ZLIB;
            else if (len >= 4) k = (b0 == 0x28 && b1 == 0xb5 && b[2] == 0x2f && b[3] == 0xfd) ? CHIP_DETECT_ZSTD : CHIP_DETECT_UNKNOWN;
        }
    }
    kind[i] = k;
}

hipError_t launch_detect(size_t n, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len, int32_t *kind,
                         hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(detect_kernel, dim3(blocks), dim3(256), 0, stream, (uint32_t)n, in_base, in_off, in_len, kind);
    return hipGetLastError();
}

}  // namespace chip
