// micro-benchmark: issue rate of integer multiply flavours on gfx950 (informational, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t seed, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77, d = b + 1234567;
    uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d;
    const uint32_t m = 0x87c37b91u;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == 1) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == 2) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(x), "+v"(y) : "v"(a), "v"(m) : "vcc");) }
        if (MODE == 3) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == 5) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == 6) { REP8(asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 2, %1" : "+v"(x), "+v"(y));) }
        if (MODE == 7) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0" : "+v"(x), "+v"(y));) }
        if (MODE == 8) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u64 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(x), "+v"(y), "+v"(a), "+v"(b) : : "vcc");) }
        if (MODE == 9) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + x + y;
}
template <int MODE> void run(const char *name, uint64_t *d)
{
    const int blocks = 256 * 8, iters = 2048;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1u, 16);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 32;
    // 256 CUs x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s at full rate
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s  (%.2f of 39.3 = full rate at 2.4 GHz)\n", name, ms, ops / ms / 1e9, ops / ms / 1e9 / 39.3);
}
int main()
{
    uint64_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    run<3>("v_add_u32 / v_xor_b32", d);
    run<9>("v_add3_u32", d);
    run<0>("v_mul_lo_u32", d);
    run<1>("v_mul_hi_u32", d);
    run<2>("v_mad_u64_u32", d);
    run<5>("v_mul_u32_u24", d);
    run<6>("v_lshl/lshr_b64", d);
    run<7>("v_lshl_add_u64", d);
    run<8>("v_cmp_lt_u64 + cndmask", d);
    return 0;
}
