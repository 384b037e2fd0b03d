// Probe (GPU box): what FETCH_SIZE reports per load for the access shapes of the inflate kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe tools/exp/fetch_probe.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/fetch_probe -o run --output-format csv -- /tmp/fetch_probe
// Three kernels over a 2 GiB buffer (far beyond L2 and the Infinity Cache), each issuing a known number of 16-byte loads per lane:
//   probe_stream   : lane-contiguous 16-byte loads (1 KiB per wave instruction) -- the shape the guide's "x 2" rule is stated for
//   probe_scatter16: every lane a random 16-byte-aligned address (one 64-byte request per lane)
//   probe_scatter1 : every lane a random byte address (the LZ77 source fetch: 16 bytes at any alignment; 23 % straddle a 64-byte line)
// The program prints the loads issued per kernel; bytes per load = FETCH_SIZE (KiB) * 1024 / loads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(256) void probe_stream(const uint8_t *buf, size_t bytes, int iters, uint32_t *sink)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
        const size_t off = ((size_t)i * nth + tid) * 16;
        acc ^= *(const u32x4 *)(buf + off % (bytes - 16) / 16 * 16);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) sink[0] = 1;
}

template <int ALIGN>
__global__ __launch_bounds__(256) void probe_scatter(const uint8_t *buf, size_t bytes, int iters, uint32_t *sink)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
        size_t off = mix(tid * 1000003ull + (uint64_t)i) % (bytes - 64);
        off = off / ALIGN * ALIGN;
        acc ^= *(const u32x4_u *)(buf + off);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) sink[0] = 1;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    uint8_t *buf;
    uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int blocks = 256 * 8, threads = 256, iters = 256;
    const double loads = (double)blocks * threads * iters;
    probe_stream<<<blocks, threads>>>(buf, bytes, iters, sink);
    hipDeviceSynchronize();
    probe_scatter<16><<<blocks, threads>>>(buf, bytes, iters, sink);
    hipDeviceSynchronize();
    probe_scatter<1><<<blocks, threads>>>(buf, bytes, iters, sink);
    hipDeviceSynchronize();
    printf("loads per kernel: %.0f (16 bytes each = %.1f MiB asked for)\n", loads, loads * 16 / 1048576.0);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
