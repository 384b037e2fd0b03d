// Probe (GPU box): issue cost of single VALU / cross-lane instructions on gfx950, in cycles per instruction per SIMD, measured with 4 waves per
// SIMD on every CU and 8 independent destination registers per wave (throughput, not latency).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/op_rate tools/exp/op_rate.hip && /tmp/op_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM)                                                                                              \
    __global__ __launch_bounds__(64) void NAME(uint32_t *o, int iters, uint32_t seed)                                \
    {                                                                                                                \
        uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 + 7, r3 = r0 ^ 5, r4 = r0 + 1, r5 = r0 + 2, r6 = r0 + 3, r7 = r0 + 4; \
        uint64_t q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;                            \
        uint32_t s = seed & 31;                                                                                      \
        for (int i = 0; i < iters; i++) {                                                                            \
            _Pragma("unroll") for (int k = 0; k < 8; k++) { ASM }                                                    \
        }                                                                                                            \
        uint32_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);     \
        if (x == 0x12345) o[threadIdx.x] = x;                                                                        \
    }

#define A1(op) asm volatile(op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8\n\t" op " %4, %4, %8\n\t" op " %5, %5, %8\n\t" op " %6, %6, %8\n\t" op " %7, %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s));
#define A3(op) asm volatile(op " %0, %0, %8, %1\n\t" op " %1, %1, %8, %2\n\t" op " %2, %2, %8, %3\n\t" op " %3, %3, %8, %4\n\t" op " %4, %4, %8, %5\n\t" op " %5, %5, %8, %6\n\t" op " %6, %6, %8, %7\n\t" op " %7, %7, %8, %0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s));
#define A64(op) asm volatile(op " %0, %8, %0\n\t" op " %1, %8, %1\n\t" op " %2, %8, %2\n\t" op " %3, %8, %3\n\t" op " %4, %8, %4\n\t" op " %5, %8, %5\n\t" op " %6, %8, %6\n\t" op " %7, %8, %7" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(s));
#define ARL(op) { uint32_t t0, t1, t2, t3; asm volatile(op " %0, %4\n\t" op " %1, %5\n\t" op " %2, %6\n\t" op " %3, %7\n\ts_nop 0" : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3)); r4 += t0; r5 += t1; r6 += t2; r7 += t3; }
#define ABP asm volatile("ds_bpermute_b32 %0, %8, %0\n\tds_bpermute_b32 %1, %8, %1\n\tds_bpermute_b32 %2, %8, %2\n\tds_bpermute_b32 %3, %8, %3\n\tds_bpermute_b32 %4, %8, %4\n\tds_bpermute_b32 %5, %8, %5\n\tds_bpermute_b32 %6, %8, %6\n\tds_bpermute_b32 %7, %8, %7\n\ts_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s));
#define ADPP asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %2, %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %4, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %5, %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %6, %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %7, %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
#define AMAD asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(s), "v"(r0) : "vcc");

#define ALIT(op) asm volatile(op " %0, 0x9e3779b9, %0\n\t" op " %1, 0x9e3779b9, %1\n\t" op " %2, 0x9e3779b9, %2\n\t" op " %3, 0x9e3779b9, %3\n\t" op " %4, 0x9e3779b9, %4\n\t" op " %5, 0x9e3779b9, %5\n\t" op " %6, 0x9e3779b9, %6\n\t" op " %7, 0x9e3779b9, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
#define AIMM(op) asm volatile(op " %0, %0, 5, 9\n\t" op " %1, %1, 5, 9\n\t" op " %2, %2, 5, 9\n\t" op " %3, %3, 5, 9\n\t" op " %4, %4, 5, 9\n\t" op " %5, %5, 5, 9\n\t" op " %6, %6, 5, 9\n\t" op " %7, %7, 5, 9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
BODY(k_add, A1("v_add_u32"))
BODY(k_add64e, A1("v_add_u32_e64"))
BODY(k_and32, A1("v_and_b32"))
BODY(k_and64e, A1("v_and_b32_e64"))
BODY(k_xor32, A1("v_xor_b32"))
BODY(k_lshr32, A1("v_lshrrev_b32"))
BODY(k_addlit, ALIT("v_add_u32"))
BODY(k_bfeimm, AIMM("v_bfe_u32"))
BODY(k_sub, A1("v_sub_u32"))
BODY(k_min, A1("v_min_u32"))
BODY(k_addf, A1("v_add_f32"))
BODY(k_and_or, A3("v_and_or_b32"))
BODY(k_lshl_or, A3("v_lshl_or_b32"))
BODY(k_lshl, A1("v_lshlrev_b32"))
BODY(k_mul, A1("v_mul_lo_u32"))
BODY(k_mulhi, A1("v_mul_hi_u32"))
BODY(k_mul24, A1("v_mul_u32_u24"))
BODY(k_alignbit, A3("v_alignbit_b32"))
BODY(k_bfe, A3("v_bfe_u32"))
BODY(k_add3, A3("v_add3_u32"))
BODY(k_lshladd, A3("v_lshl_add_u32"))
BODY(k_perm, A3("v_perm_b32"))
BODY(k_mad24, A3("v_mad_u32_u24"))
BODY(k_shr64, A64("v_lshrrev_b64"))
BODY(k_shl64, A64("v_lshlrev_b64"))
BODY(k_rfl, ARL("v_readfirstlane_b32"))
BODY(k_bperm, ABP)
BODY(k_dpp, ADPP)
BODY(k_mad64, AMAD)

template <class K>
void run(const char *name, K kern, int per_iter)
{
    uint32_t *o;
    hipMalloc(&o, 4096);
    const int iters = 4000, wps = 4;
    dim3 grid(256 * 4 * wps), block(64);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kern<<<grid, block>>>(o, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<grid, block>>>(o, iters, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double per_simd = (double)iters * 8 * per_iter * wps;  // instructions issued per SIMD
    printf("%-22s %.3f ms  %.2f ns per instruction per SIMD  (= %.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(o);
}
int main()
{
    run("v_add_u32", k_add, 8);
    run("v_add_u32_e64", k_add64e, 8);
    run("v_and_b32 (e32)", k_and32, 8);
    run("v_and_b32_e64", k_and64e, 8);
    run("v_xor_b32 (e32)", k_xor32, 8);
    run("v_lshrrev_b32 (e32)", k_lshr32, 8);
    run("v_sub_u32", k_sub, 8);
    run("v_min_u32", k_min, 8);
    run("v_add_f32", k_addf, 8);
    run("v_add_u32 + literal", k_addlit, 8);
    run("v_bfe_u32 imm", k_bfeimm, 8);
    run("v_and_or_b32", k_and_or, 8);
    run("v_lshl_or_b32", k_lshl_or, 8);
    run("v_lshlrev_b32", k_lshl, 8);
    run("v_mul_lo_u32", k_mul, 8);
    run("v_mul_hi_u32", k_mulhi, 8);
    run("v_mul_u32_u24", k_mul24, 8);
    run("v_mad_u32_u24", k_mad24, 8);
    run("v_alignbit_b32", k_alignbit, 8);
    run("v_bfe_u32", k_bfe, 8);
    run("v_add3_u32", k_add3, 8);
    run("v_lshl_add_u32", k_lshladd, 8);
    run("v_perm_b32", k_perm, 8);
    run("v_lshrrev_b64", k_shr64, 8);
    run("v_lshlrev_b64", k_shl64, 8);
    run("v_mad_u64_u32", k_mad64, 8);
    run("v_readfirstlane_b32", k_rfl, 4);
    run("ds_bpermute_b32", k_bperm, 8);
    run("v_add_u32_dpp row_shr", k_dpp, 8);
    return 0;
}
