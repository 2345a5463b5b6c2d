// micro-benchmark: VALU issue rates on gfx950, in lane-operations per second, for the instruction classes the VALU-bound kernels of
// the path are made of (scan_kernel: MurmurHash3 = 64-bit multiplies built from v_mul_lo/hi/mad, xor-shifts, 64-bit shifts, compares;
// combine_kernel: the roll = 64-bit shifts / funnel shifts, bit reversal, 64-bit compare + select).  Informational, not part of the product;
// its output is committed as profiles/rNN_valu_rates.txt and tools/valu_floor.py prices a kernel's ISA mix with it.
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp/mulrate tools/exp/mulrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define REP8(x) x x x x x x x x
enum { ADDXOR, ADD3, MUL_LO, MUL_HI, MAD64, MUL24, SH64, LSHLADD64, CMP64_CND, SH32, ALIGNBIT, BFE, ANDOR, PERM, CNDMASK, BFREV, MINU32, CMP32_CND, DPP_MOV, DPP_ADD, MOV, LSHL_OR, XOR3ISH, ADDC64, ANDOR2, SUB, LSHLADD32, CMP_ONLY, CMP_CND2, CMP_CND4, CND_SGPR, CMP64_CND2, MOV64, READLANE, NOT, OR3, BITOP3, FFBL, MBCNT, SHL32, SHR32, MIN64EMU, CMP_CND3_SGPR, CMP_CND3_SPACED, MIN_F64, NMODES };
static const char *names[NMODES] = {"v_add_u32 / v_xor_b32", "v_add3_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_lshlrev_b64 / v_lshrrev_b64",
    "v_lshl_add_u64", "v_cmp_lt_u64 + v_cndmask_b32", "v_lshlrev_b32 / v_lshrrev_b32", "v_alignbit_b32", "v_bfe_u32", "v_and_or_b32", "v_perm_b32", "v_cndmask_b32 (vcc fixed)",
    "v_bfrev_b32", "v_min_u32", "v_cmp_lt_u32 + v_cndmask_b32", "v_mov_b32 dpp row_shr:1", "v_add_u32 dpp row_shr:1", "v_mov_b32", "v_lshl_or_b32", "v_xad_u32", "v_add_co_u32 + v_addc_co_u32",
    "v_and_b32 / v_or_b32", "v_sub_u32", "v_lshl_add_u32", "v_cmp_lt_u32 (vcc) alone", "v_cmp_lt_u32 + 2 v_cndmask_b32", "v_cmp_lt_u32 + 3 v_cndmask_b32 (vcc)", "v_cndmask_b32_e64 (sgpr mask fixed)", "v_cmp_lt_u64 + 2 v_cndmask_b32 (64-bit min)",
    "v_mov_b64", "v_readlane_b32 / v_writelane_b32", "v_not_b32", "v_or3_b32", "v_bitop3_b32", "v_ffbl_b32", "v_mbcnt_lo + v_mbcnt_hi", "v_lshlrev_b32 alone", "v_lshrrev_b32 alone", "64-bit min as sub/subb + 2 cndmask (per 4)", "v_cmp_lt_u32_e64 s[20:21] + 3 v_cndmask_b32_e64", "v_cmp_lt_u32 + (v_cndmask, v_add) x 3 (vcc)", "v_min_f64 (64-bit minimum of sign-free, NaN-free bit patterns)"};
// class key used by tools/valu_floor.py to map ISA mnemonics onto a measured rate
static const char *keys[NMODES] = {"add32", "add3", "mul_lo", "mul_hi", "mad64", "mul24", "shift64", "lshl_add64", "cmp64_cnd", "shift32", "alignbit", "bfe", "and_or", "perm", "cndmask",
    "bfrev", "min32", "cmp32_cnd", "dpp_mov", "dpp_add", "mov", "lshl_or", "xad", "addc64",
    "and_or32", "sub32", "lshl_add32", "cmp32", "cmp32_cnd2", "cmp32_cnd4", "cnd_sgpr", "cmp64_cnd2", "mov64", "lane", "not", "or3", "bitop3", "ffbl", "mbcnt", "shl32", "shr32", "min64emu", "cmp32_cnd3_sgpr", "cmp32_cnd3_spaced", "min_f64"};

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t seed, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77, d = b + 1234567;
    uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d;
    const uint32_t m = 0x87c37b91u;
    for (int i = 0; i < iters; ++i) {
        if (MODE == MUL_LO) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == MUL_HI) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == MAD64) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(x), "+v"(y) : "v"(a), "v"(m) : "vcc");) }
        if (MODE == ADDXOR) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == MUL24) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == SH64) { REP8(asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 2, %1" : "+v"(x), "+v"(y));) }
        if (MODE == LSHLADD64) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0" : "+v"(x), "+v"(y));) }
        if (MODE == CMP64_CND) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u64 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(x), "+v"(y), "+v"(a), "+v"(b) : : "vcc");) }
        if (MODE == ADD3) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == SH32) { REP8(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 5, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == ALIGNBIT) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 9\n v_alignbit_b32 %2, %2, %3, 11\n v_alignbit_b32 %3, %3, %0, 13" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == BFE) { REP8(asm volatile("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 2, 30\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 2, 30" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == ANDOR) { REP8(asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == PERM) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0x00010203u));) }
        if (MODE == CNDMASK) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (MODE == BFREV) { REP8(asm volatile("v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == MINU32) { REP8(asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == CMP32_CND) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (MODE == DPP_MOV) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == DPP_ADD) { REP8(asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == MOV) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == LSHL_OR) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 5, %2\n v_lshl_or_b32 %2, %2, 7, %3\n v_lshl_or_b32 %3, %3, 9, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == XOR3ISH) { REP8(asm volatile("v_xad_u32 %0, %0, %1, %2\n v_xad_u32 %1, %1, %2, %3\n v_xad_u32 %2, %2, %3, %0\n v_xad_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == ANDOR2) { REP8(asm volatile("v_and_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == SUB) { REP8(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == LSHLADD32) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 5, %2\n v_lshl_add_u32 %2, %2, 7, %3\n v_lshl_add_u32 %3, %3, 9, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == CMP_ONLY) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (MODE == CMP_CND2) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n v_add_u32 %0, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (MODE == CMP_CND4) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (MODE == CND_SGPR) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");) }
        if (MODE == CMP64_CND2) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_lshl_add_u64 %1, %1, 0, %0" : "+v"(x), "+v"(y), "+v"(a), "+v"(b) : : "vcc");) }
        if (MODE == MOV64) { REP8(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n v_mov_b64 %0, %1\n v_mov_b64 %1, %0" : "+v"(x), "+v"(y));) }
        if (MODE == READLANE) { REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %1, s20, 5\n v_readlane_b32 s21, %2, 7\n v_writelane_b32 %3, s21, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");) }
        if (MODE == NOT) { REP8(asm volatile("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == OR3) { REP8(asm volatile("v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == BITOP3) { REP8(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == FFBL) { REP8(asm volatile("v_ffbl_b32 %0, %1\n v_ffbl_b32 %1, %2\n v_ffbl_b32 %2, %3\n v_ffbl_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == MBCNT) { REP8(asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n v_mbcnt_hi_u32_b32 %0, %2, %0\n v_mbcnt_lo_u32_b32 %1, %3, 0\n v_mbcnt_hi_u32_b32 %1, %2, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == SHL32) { REP8(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 5, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == SHR32) { REP8(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 5, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == MIN64EMU) { REP8(asm volatile("v_sub_co_u32 %4, vcc, %0, %2\n v_subb_co_u32 %4, vcc, %1, %3, vcc\n v_cndmask_b32 %0, %2, %0, vcc\n v_cndmask_b32 %1, %3, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");) }
        if (MODE == CMP_CND3_SGPR) { REP8(asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");) }
        if (MODE == CMP_CND3_SPACED) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_add_u32 %1, %1, %4\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");) }
        if (MODE == MIN_F64) { REP8(asm volatile("v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %0\n v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %0" : "+v"(x), "+v"(y));) }
        if (MODE == ADDC64) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc\n v_add_co_u32 %2, vcc, %2, %0\n v_addc_co_u32 %3, vcc, %3, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + x + y;
}

// blocks_per_cu workgroups of 256 threads per CU = that many waves per SIMD
template <int MODE> static double run(uint64_t *d, int blocks_per_cu, int ncu)
{
    const int blocks = ncu * blocks_per_cu, iters = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1u, 16);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double ops = (double)blocks * 256 * iters * 32;
    return ops / best / 1e9;                 // T lane-ops/s
}
template <int MODE> static void row(uint64_t *d, int ncu, double clk_ghz)
{
    const double full32 = ncu * 4 * 32 * clk_ghz * 1e-3;       // SIMD-32: one wave-instruction per 2 cycles
    const double r8 = run<MODE>(d, 8, ncu), r4 = run<MODE>(d, 4, ncu), r1 = run<MODE>(d, 1, ncu);
    printf("%-34s key=%-10s  8 waves/SIMD %7.2f  4 waves/SIMD %7.2f  1 wave/SIMD %7.2f  T lane-ops/s   (%.3f / %.3f / %.3f of %.1f; cycles per wave-instruction per SIMD at 8 waves: %.2f)\n",
           names[MODE], keys[MODE], r8, r4, r1, r8 / full32, r4 / full32, r1 / full32, full32, 64.0 * ncu * 4 * clk_ghz * 1e-3 / r8);
}
template <int MODE> static void all(uint64_t *d, int ncu, double clk) { row<MODE>(d, ncu, clk); if constexpr (MODE + 1 < NMODES) all<MODE + 1>(d, ncu, clk); }

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount; const double clk = p.clockRate * 1e-6;
    printf("# %s: %d CUs, clockRate %.3f GHz; a lane-op = one lane of one VALU instruction; 32 independent-ish instructions per loop iteration, 4096 iterations, best of 3\n", p.gcnArchName, ncu, clk);
    printf("# ceilings: SIMD-32 (one wave64 instruction per 2 cycles per SIMD) = %.1f T lane-ops/s; SIMD-16 (4 cycles) = %.1f\n", ncu * 4 * 32 * clk * 1e-3, ncu * 4 * 16 * clk * 1e-3);
    uint64_t *d; (void)hipMalloc(&d, (size_t)ncu * 8 * 256 * 8);
    all<0>(d, ncu, clk);
    return 0;
}
