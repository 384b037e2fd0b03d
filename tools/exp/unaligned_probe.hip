// Probe (GPU box): do unaligned 16-byte global loads, unaligned LDS dword/short stores and loads work on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
struct __attribute__((packed)) U32u { uint32_t v; };
struct __attribute__((packed)) U16u { uint16_t v; };
struct __attribute__((packed)) U128u { uint32_t x, y, z, w; };
__global__ void k(const uint8_t *g, uint8_t *o, int step)
{
    __shared__ uint8_t s[8192];
    int i = threadIdx.x;
    for (int j = i; j < 8192; j += 64) s[j] = 0;
    __syncthreads();
    int so = i * step;           // unaligned source offsets
    U128u v = *(const U128u *)(g + so);
    int d = i * 19 + 1;          // unaligned LDS destinations
    ((U32u *)(s + d))->v = v.x;
    ((U32u *)(s + d + 4))->v = v.y;
    ((U16u *)(s + d + 8))->v = (uint16_t)v.z;
    s[d + 10] = (uint8_t)(v.z >> 16);
    __syncthreads();
    U32u r0 = *(U32u *)(s + d + 1), r1 = *(U32u *)(s + d + 5);
    ((U32u *)(o + i * 11))->v = r0.v;         // unaligned global store
    ((U32u *)(o + i * 11 + 4))->v = r1.v;
}
int main()
{
    uint8_t h[4096], ho[2048], *g, *o;
    for (int i = 0; i < 4096; i++) h[i] = (uint8_t)(i * 7 + 3);
    hipMalloc(&g, 4096); hipMalloc(&o, 2048);
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    int bad = 0;
    for (int step = 1; step <= 37; step += 3) {
        hipMemset(o, 0, 2048);
        k<<<1, 64>>>(g, o, step);
        if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
        hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
        // lanes overlap in o (11-byte stride, 8 bytes written): check lane 63 fully and each lane's first 8 bytes that the next lane does not overwrite (first 11)
        for (int i = 0; i < 64; i++)
            for (int b = 0; b < 8; b++) {
                if (i < 63 && b >= 11) continue;
                uint8_t want = h[i * step + 1 + b];
                if (ho[i * 11 + b] != want) bad++;
            }
    }
    printf("unaligned probe: %s (%d bad bytes)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
