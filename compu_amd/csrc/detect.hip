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

// One lane per unit: route by Detection::detect.  Units that are neither gzip/zlib nor zstd are answered here, as the
// decoders would: too short to tell -> NeedInput (detect() == None), anything else -> CHIP_UNKNOWN_FORMAT.
__global__ void route_kernel(BatchArgs a, uint32_t *sel_inflate, uint32_t *sel_zstd, uint32_t *counts)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int32_t kind = detect_kind(a.in_base + a.in_off[i], a.in_len[i]);
    if (kind == CHIP_DETECT_GZIP || kind == CHIP_DETECT_ZLIB) {
        sel_inflate[atomicAdd(&counts[0], 1u)] = i;
    } else if (kind == CHIP_DETECT_ZSTD) {
        sel_zstd[atomicAdd(&counts[1], 1u)] = i;
    } else {
        a.out_len[i] = 0;
        a.in_used[i] = 0;
        a.status[i] = kind == CHIP_DETECT_NONE ? CHIP_NEED_INPUT : CHIP_UNKNOWN_FORMAT;
    }
}

hipError_t launch_route(const BatchArgs &a, uint32_t *sel_inflate, uint32_t *sel_zstd, uint32_t *counts, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(counts, 0, 8, stream);
    if (e != hipSuccess) return e;
    uint32_t blocks = (a.n + 255u) / 256u;
    hipLaunchKernelGGL(route_kernel, dim3(blocks), dim3(256), 0, stream, a, sel_inflate, sel_zstd, counts);
    return hipGetLastError();
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
