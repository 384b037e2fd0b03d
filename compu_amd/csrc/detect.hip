// Batched form of Detection::detect (src/decoder/mod.rs:28-114): classify each unit by its first
// 2-4 bytes.  One lane per unit; the router in front of a mixed gzip+zstd batch.
#include "chip_internal.h"

namespace chip {

__global__ void detect_kernel(uint32_t n, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len, int32_t *kind)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kind[i] = detect_kind(in_base + in_off[i], in_len[i]);
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
