// hsk_device.h -- device-side building blocks shared by the kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsk {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

constexpr int WAVE = 64;

// ---- MurmurHash3_x64_128(seed 313) low word of one 8-byte key --------------------------------
// Closed form of reference src/hashfuncs.cpp:42-114 for len == 8 (no body block, tail case 8),
// as called by Mmer::GetHash (include/supermer.hpp:308) through murmurhash3_64 (:233).
__host__ __device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ __forceinline__ u64 fmix64(u64 k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
__host__ __device__ __forceinline__ u64 murmur64_8(u64 key)
{
    u64 k1 = key * 0x87c37b91114253d5ULL;
    k1 = rotl64(k1, 31);
    k1 *= 0x4cf5ad432745937fULL;
    u64 h1 = 313ULL ^ k1, h2 = 313ULL;
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    return h1 + h2;
}

// The same hash over NW 64-bit words (16 / 24 bytes: Mmer<2> / Mmer<3>, minimizers of 33..63 / 65..94 bases): one body
// block of 16 bytes, and for 24 bytes an 8-byte tail (reference src/hashfuncs.cpp:42-114 with len = 8 * NW).
template <int NW>
__host__ __device__ __forceinline__ u64 murmur64_words(const u64 (&w)[NW])
{
    if (NW == 1) return murmur64_8(w[0]);
    const u64 c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    u64 h1 = 313ULL, h2 = 313ULL;
    u64 k1 = w[0], k2 = w[NW > 1 ? 1 : 0];
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
    if (NW == 3) { u64 t1 = w[NW > 2 ? 2 : 0]; t1 *= c1; t1 = rotl64(t1, 31); t1 *= c2; h1 ^= t1; }
    h1 ^= (u64)(8 * NW); h2 ^= (u64)(8 * NW);
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    return h1 + h2;
}

// ---- 2-bit sequence helpers ------------------------------------------------------------------
// Reverse the order of the 32 2-bit groups of x (base i <-> base 31-i).
__host__ __device__ __forceinline__ u64 rev2(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((x & 0x0f0f0f0f0f0f0f0fULL) << 4);
    x = __builtin_bswap64(x);
#endif
    // full bit reversal swapped the two bits inside every base: swap them back
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

// Reverse complement of an M-mer (M <= 32) held left-aligned in one word
// (Mmer::GetTwin, include/supermer.hpp:266-296 for MLONGS == 1).
__host__ __device__ __forceinline__ u64 twin1(u64 fw, int len)
{
    u64 t = ~rev2(fw);
    return t << (64 - 2 * len);   // len in [1,31]; garbage of the complement is shifted out
}

// Multi-word k-mer, left-aligned: base i in w[i/32] at shift 2*(31 - i%32) (kmer.hpp:166-186).
template <int NW> struct Mer { u64 w[NW]; };

// Kmer::GetTwin (kmer.hpp:266-296): reverse complement of a K-mer spanning NW words.
template <int NW>
__host__ __device__ __forceinline__ Mer<NW> twin(const Mer<NW> &m, int k)
{
    Mer<NW> t;
#pragma unroll
    for (int l = 0; l < NW; ++l) t.w[NW - 1 - l] = ~rev2(m.w[l]);
    const int sh = (NW * 64 - 2 * k);          // unused low bits of the last word; 0 < sh < 64
    if (sh) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            u64 nxt = (i + 1 < NW) ? t.w[i + 1] : 0;
            t.w[i] = (t.w[i] << sh) | (nxt >> (64 - sh));
        }
    }
    return t;
}

// Kmer::operator< (kmer.hpp:217-229): word 0 is compared first.
template <int NW>
__host__ __device__ __forceinline__ bool mer_less(const Mer<NW> &a, const Mer<NW> &b)
{
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        if (a.w[i] < b.w[i]) return true;
        if (a.w[i] > b.w[i]) return false;
    }
    return false;
}

// Kmer::GetRep (kmer.hpp:299-303)
template <int NW>
__host__ __device__ __forceinline__ Mer<NW> canonical(const Mer<NW> &m, int k)
{
    Mer<NW> t = twin<NW>(m, k);
    return mer_less<NW>(t, m) ? t : m;
}

// 64 bits of a big-endian bit stream stored as byte-swapped 32-bit words (word j holds stream
// bits [32j, 32j+32) with the first bit in the MSB), starting at bit `bit`.
__device__ __forceinline__ u64 bits64_be32(const u32 *words, u32 bit)
{
    u32 i = bit >> 5, s = bit & 31;
    u64 hi = ((u64)words[i] << 32) | words[i + 1];
    u64 lo = (u64)words[i + 2] << 32;
    return s ? ((hi << s) | (lo >> (64 - s))) : hi;
}

// Same from global memory bytes: the stream is the byte array itself (first base in the MSBs of
// byte 0).  p8 must be 8-byte aligned; reads two aligned 64-bit words around bit `bit`.
__device__ __forceinline__ u64 bits64_bytes(const u64 *p8, u64 bit)
{
    u64 i = bit >> 6; u32 s = (u32)(bit & 63);
    u64 hi = __builtin_bswap64(p8[i]);
    if (!s) return hi;
    u64 lo = __builtin_bswap64(p8[i + 1]);
    return (hi << s) | (lo >> (64 - s));
}
// same, never reading a word at or beyond index `nwords` (bits past the buffer read as zero)
__device__ __forceinline__ u64 bits64_bytes_clamped(const u64 *p8, u64 bit, u64 nwords)
{
    u64 i = bit >> 6; u32 s = (u32)(bit & 63);
    u64 hi = i < nwords ? __builtin_bswap64(p8[i]) : 0;
    if (!s) return hi;
    u64 lo = (i + 1 < nwords) ? __builtin_bswap64(p8[i + 1]) : 0;
    return (hi << s) | (lo >> (64 - s));
}

// ---- hash % ntasks without a 64-bit division -------------------------------------------------------
// (GetMinimizerOwner, reference src/kmerops.cpp:1044: hash % tot_tasks; must be the exact remainder).
// For d <= 1024 the 64-bit value is folded 16 bits at a time:  x = a*2^48 + b*2^32 + c*2^16 + e  =>
// x mod d = (a*(2^48 mod d) + b*(2^32 mod d) + c*(2^16 mod d) + e) mod d, the sum stays below 2^28, so
// three full-rate 24-bit multiply-adds and ONE Barrett step (floor(2^32/d)) replace three 64-bit fastmods.
struct FastMod { u32 d; u32 c16, c32, c48; u32 inv; };
inline FastMod make_fastmod(u32 d)
{
    FastMod f; f.d = d;
    f.c16 = d > 1 ? (u32)((1ULL << 16) % d) : 0; f.c32 = d > 1 ? (u32)((1ULL << 32) % d) : 0; f.c48 = d > 1 ? (u32)((1ULL << 48) % d) : 0;
    f.inv = d > 1 ? (u32)((1ULL << 32) / d) : 0;
    return f;
}
__host__ __device__ __forceinline__ u32 fastmod64(u64 x, const FastMod &f)
{
    if (f.d <= 1) return 0;
    const u32 a = (u32)(x >> 48), b = (u32)(x >> 32) & 0xFFFFu, c = (u32)(x >> 16) & 0xFFFFu, e = (u32)x & 0xFFFFu;
    const u32 t = a * f.c48 + b * f.c32 + c * f.c16 + e;            // < 3 * 2^16 * 2^10 + 2^16 < 2^28
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 q = __umulhi(t, f.inv);
#else
    const u32 q = (u32)(((u64)t * f.inv) >> 32);
#endif
    u32 r = t - q * f.d;                                             // Barrett: q is the quotient or one less
    if (r >= f.d) r -= f.d;
    if (r >= f.d) r -= f.d;
    return r;
}

// ---- wave / block scans ------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }

// 32-bit sums: six adds whose operand comes through the data-parallel-primitive path of the VALU (row shifts inside the four
// rows of 16 lanes, then the last lane of row 0 / 2 into row 1 / 3 and lane 31 into rows 2 and 3): no LDS permutes, no selects
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1 (lanes without a source add the `old` operand: 0)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15, rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31, rows 2 and 3
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v)
{
    if constexpr (sizeof(T) == 4 && (T)(-1) > (T)0) return (T)wave_incl_scan_dpp((u32)v);
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        T n = __shfl_up(v, o, WAVE);
        if (lane >= o) v += n;
    }
    return v;
}

// Exclusive block scan over 256 threads (4 waves); `scratch` has >= 8 T's in LDS.
// Returns the exclusive prefix of v; *total gets the block total.
template <typename T>
__device__ __forceinline__ T block_excl_scan_256(T v, T *scratch, T *total)
{
    const int lane = lane_id(), w = threadIdx.x >> 6;
    T inc = wave_incl_scan(v);
    if (lane == WAVE - 1) scratch[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { T s = scratch[i]; if (i < w) base += s; tot += s; }
    __syncthreads();
    if (total) *total = tot;
    return base + inc - v;
}

// Workgroup barrier for LDS traffic only: waits for the wave's LDS operations, not for its outstanding global loads, stores
// and atomics.  __syncthreads() drains those too, which puts every prefetch issued before it (and the acknowledgement of
// every store) on the critical path.  Use where the waves hand data to each other through LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ T block_excl_scan_256_lds(T v, T *scratch, T *total)      // block_excl_scan_256 with lds_barrier()
{
    const int lane = lane_id(), w = threadIdx.x >> 6;
    T inc = wave_incl_scan(v);
    if (lane == WAVE - 1) scratch[w] = inc;
    lds_barrier();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { T s = scratch[i]; if (i < w) base += s; tot += s; }
    lds_barrier();
    if (total) *total = tot;
    return base + inc - v;
}

// ---- radix digit plan (hsk_sort.h; hsk_expand.h builds the digit histograms while it writes the keys) ----
constexpr int MAX_PASSES = 24;
struct PassDesc { int word; int shift; int bits; };

// k[word] without dynamic register indexing (which would spill the key array to scratch)
template <int NW> __device__ __forceinline__ u64 pick_word(const u64 *k, int word)
{
    if (NW == 1) return k[0];
    if (NW == 2) return word == 0 ? k[0] : k[1];
    return word == 0 ? k[0] : (word == 1 ? k[1] : k[NW - 1]);
}

// splitmix64 -- the synthetic-read generator's PRNG (also in hysortk_amd/synth.py)
__host__ __device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

} // namespace hsk
