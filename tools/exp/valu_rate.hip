// Probe (GPU box): VALU issue rate per SIMD for integer ops at 1..8 waves per SIMD, dependent chains vs independent.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int ILP>
__global__ void k(uint32_t *o, int iters, uint32_t seed)
{
    uint32_t a[ILP];
#pragma unroll
    for (int j = 0; j < ILP; j++) a[j] = seed + threadIdx.x + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int j = 0; j < ILP; j++) a[j] = (a[j] << 1) ^ (a[j] + 0x9e3779b9u);  // 2-3 VALU, dependent per chain
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= a[j];
    if (s == 0x12345) o[threadIdx.x] = s;
}
template <int ILP>
void run(int wps)
{
    uint32_t *o;
    hipMalloc(&o, 4096);
    const int iters = 20000;
    // one block = 64 threads = 1 wave; grid = 256 CUs * 4 SIMDs * wps waves
    dim3 grid(256 * 4 * wps), block(64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<ILP><<<grid, block>>>(o, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<ILP><<<grid, block>>>(o, iters, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)iters * 16 * ILP * wps;  // chain steps per SIMD
    printf("ILP %d waves/SIMD %d: %.3f ms, %.2f ns per chain-step per SIMD (x clock = cycles)\n", ILP, wps, ms, ms * 1e6 / ops);
    hipFree(o);
}
int main()
{
    for (int w : {1, 2, 4, 6, 8}) run<1>(w);
    for (int w : {1, 2, 4, 8}) run<4>(w);
    return 0;
}
