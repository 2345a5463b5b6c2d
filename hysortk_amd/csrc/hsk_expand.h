// hsk_expand.h -- supermers -> canonical k-mer records of one task (device).
//
// Replaces GatheredSupermer::receive_from_buffer_stage2 (reference src/kmerops.cpp:484-521) and
// what it calls: DnaSeq view + TKmer::GetRepKmers (include/kmer.hpp:314-341).  The reference
// rolls GetExtension along the supermer and recomputes GetTwin per k-mer; here every k-mer is an
// independent lane: it pulls its 2K bits straight out of the re-aligned supermer bytes at bit
// offset 2*i and canonicalises with a bit-reversal (no table, no rolling state), so writes of
// consecutive lanes are consecutive records (coalesced).
//
// A task's supermers arrive as `nseg` segments (one per source rank; one on a single GPU).  Three
// launches: tile sums (bytes, k-mers per 2048 supermers) -> per-segment exclusive scan -> expand.
#pragma once
#include <vector>
#include "hsk_device.h"

namespace hsk {

constexpr int EXP_THREADS = 256;
constexpr int EXP_SPT = 8;
constexpr int EXP_TILE = EXP_THREADS * EXP_SPT;   // supermers per tile
constexpr int EXP_CHUNK = 2048;                   // output slots produced per step inside a tile

struct ExpSeg {
    u64 sup_off;     // first supermer slot (index into sm_len / sm_pos / sm_rid)
    u64 n_sup;
    u64 byte_off;    // first byte in sm_bytes
    u64 kmer_off;    // first output record of this segment, relative to the task's key array
    u64 tile_start;  // index of this segment's first tile in the task's tile list
};

// host-side description of one task's input: its segments, tile count and k-mer total
struct TaskSegs { std::vector<ExpSeg> segs; u64 ntiles = 0; u64 nkmers = 0; };

__device__ __forceinline__ int seg_of_tile(const ExpSeg *segs, int nseg, u64 tile)
{
    int s = 0;
    while (s + 1 < nseg && segs[s + 1].tile_start <= tile) ++s;
    return s;
}

// tile_sum[tile] = {bytes, kmers}
__global__ __launch_bounds__(EXP_THREADS) void expand_tilesum_kernel(const ExpSeg *segs, int nseg, const u8 *sm_len, int k, u64 *tile_sum)
{
    __shared__ u64 s_red[2 * 4];
    const u64 tile = blockIdx.x;
    const int sg = seg_of_tile(segs, nseg, tile);
    const ExpSeg seg = segs[sg];
    const u64 first = (tile - seg.tile_start) * EXP_TILE;
    u64 nb = 0, nk = 0;
    for (int i = 0; i < EXP_SPT; ++i) {
        u64 s = first + (u64)i * EXP_THREADS + threadIdx.x;
        if (s < seg.n_sup) { u32 len = sm_len[seg.sup_off + s]; nb += (len + 3) >> 2; nk += len - k + 1; }
    }
    for (int o = 32; o > 0; o >>= 1) { nb += __shfl_down(nb, o, WAVE); nk += __shfl_down(nk, o, WAVE); }
    if ((threadIdx.x & 63) == 0) { s_red[2 * (threadIdx.x >> 6)] = nb; s_red[2 * (threadIdx.x >> 6) + 1] = nk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 b = 0, kk = 0;
        for (int w = 0; w < 4; ++w) { b += s_red[2 * w]; kk += s_red[2 * w + 1]; }
        tile_sum[2 * tile] = b; tile_sum[2 * tile + 1] = kk;
    }
}

// one block per segment: tile_off[tile] = {absolute byte offset, k-mer offset relative to task}
__global__ __launch_bounds__(EXP_THREADS) void expand_scan_kernel(const ExpSeg *segs, int nseg, u64 ntiles_total, const u64 *tile_sum, u64 *tile_off)
{
    __shared__ u64 s_scr[8];
    __shared__ u64 s_carry[2];
    const int sg = blockIdx.x;
    const ExpSeg seg = segs[sg];
    const u64 t0 = seg.tile_start;
    const u64 t1 = (sg + 1 < nseg) ? segs[sg + 1].tile_start : ntiles_total;
    if (threadIdx.x == 0) { s_carry[0] = seg.byte_off; s_carry[1] = seg.kmer_off; }
    __syncthreads();
    for (u64 base = t0; base < t1; base += EXP_THREADS) {
        const u64 t = base + threadIdx.x;
        u64 b = 0, kk = 0;
        if (t < t1) { b = tile_sum[2 * t]; kk = tile_sum[2 * t + 1]; }
        u64 tb, tk;
        u64 eb = block_excl_scan_256<u64>(b, s_scr, &tb);
        u64 ek = block_excl_scan_256<u64>(kk, s_scr, &tk);
        if (t < t1) { tile_off[2 * t] = s_carry[0] + eb; tile_off[2 * t + 1] = s_carry[1] + ek; }
        __syncthreads();
        if (threadIdx.x == 0) { s_carry[0] += tb; s_carry[1] += tk; }
        __syncthreads();
    }
}

// Source of the bases: either the re-aligned byte stream of received supermers (sm_gpos == null;
// byte offsets come from the prefix sums) or, for supermers that never left this GPU, the rank's
// packed reads themselves (sm_gpos[s] = position of the supermer's first base; `src8` = packed reads
// rounded down to 8 bytes, `src_bit0` = bit offset of the first read base inside src8).
template <int NW, bool EXT>
__global__ __launch_bounds__(EXP_THREADS) void expand_kernel(const ExpSeg *segs, int nseg, const u8 *sm_len, const u64 *src8, u64 src_bit0, u64 src_words,
                                                              const u64 *sm_gpos, const u32 *sm_pos, const int32_t *sm_rid, const u64 *tile_off,
                                                              int k, u64 *keys_out, u64 *vals_out)
{
    __shared__ u32 s_boff[EXP_TILE + 1];
    __shared__ u32 s_koff[EXP_TILE + 1];
    __shared__ u32 s_scr[8];
    __shared__ u64 s_mask[EXP_CHUNK / 64];
    __shared__ u32 s_wpre[EXP_CHUNK / 64];
    const u64 tile = blockIdx.x;
    const int sg = seg_of_tile(segs, nseg, tile);
    const ExpSeg seg = segs[sg];
    const u64 first = (tile - seg.tile_start) * EXP_TILE;
    const u32 ns = (u32)((seg.n_sup - first) < (u64)EXP_TILE ? (seg.n_sup - first) : (u64)EXP_TILE);
    const int tid = threadIdx.x;

    // blocked arrangement: thread t owns supermers [t*SPT, t*SPT+SPT) of the tile
    u32 nb[EXP_SPT], nk[EXP_SPT], sb = 0, sk = 0;
#pragma unroll
    for (int i = 0; i < EXP_SPT; ++i) {
        u32 s = tid * EXP_SPT + i;
        u32 len = (s < ns) ? sm_len[seg.sup_off + first + s] : 0;
        nb[i] = (s < ns) ? ((len + 3) >> 2) : 0;
        nk[i] = (s < ns) ? (len - k + 1) : 0;
        sb += nb[i]; sk += nk[i];
    }
    u32 totb, totk;
    u32 eb = block_excl_scan_256<u32>(sb, s_scr, &totb);
    u32 ek = block_excl_scan_256<u32>(sk, s_scr, &totk);
#pragma unroll
    for (int i = 0; i < EXP_SPT; ++i) {
        s_boff[tid * EXP_SPT + i] = eb; s_koff[tid * EXP_SPT + i] = ek;
        eb += nb[i]; ek += nk[i];
    }
    if (tid == EXP_THREADS - 1) { s_boff[EXP_TILE] = eb; s_koff[EXP_TILE] = ek; }
    __syncthreads();

    const u64 byte_abs = tile_off[2 * tile];
    const u64 kbase = tile_off[2 * tile + 1];
    const u64 lastmask = ~0ULL << (64 * NW - 2 * k);      // 0 < 64*NW - 2k < 64 (k % 32 != 0)

    // The tile's k-mers are produced in chunks of 2048 consecutive output slots.  Which supermer a slot belongs to
    // is NOT searched per k-mer: the supermers that start inside the chunk set one bit each in a 2048-bit mask, and
    // slot j belongs to supermer  S0 + popcount(mask bits <= j) - 1  (S0 = supermers starting before the chunk: one
    // search per chunk).  Lane t handles slots t, t+256, ...: consecutive lanes write consecutive records.
    const int lane = tid & 63;
    for (u32 c0 = 0; c0 < totk; c0 += EXP_CHUNK) {
        const u32 c1 = (c0 + EXP_CHUNK < totk) ? c0 + EXP_CHUNK : totk;
        // supermers starting in [c0, c1): a contiguous index range [sa, sb)
        u32 sa, sb;
        { u32 lo = 0, hi = ns; while (lo < hi) { u32 mid = (lo + hi) >> 1; if (s_koff[mid] < c0) lo = mid + 1; else hi = mid; } sa = lo; }
        { u32 lo = sa, hi = ns; while (lo < hi) { u32 mid = (lo + hi) >> 1; if (s_koff[mid] < c1) lo = mid + 1; else hi = mid; } sb = lo; }
        if (tid < EXP_CHUNK / 64) s_mask[tid] = 0;
        __syncthreads();
        for (u32 sidx = sa + tid; sidx < sb; sidx += EXP_THREADS) {
            const u32 b = s_koff[sidx] - c0;
            atomicOr((unsigned long long *)&s_mask[b >> 6], 1ULL << (b & 63));
        }
        __syncthreads();
        if (tid < 64) {
            const u32 v = tid < EXP_CHUNK / 64 ? (u32)__popcll(s_mask[tid]) : 0;
            const u32 inc = wave_incl_scan<u32>(v);
            if (tid < EXP_CHUNK / 64) s_wpre[tid] = inc - v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < EXP_CHUNK / EXP_THREADS; ++r) {
            const u32 j = c0 + r * EXP_THREADS + tid;
            if (j >= c1) continue;
            const u32 b = j - c0;
            const u32 w = b >> 6;
            const u64 below = ((b & 63) == 63) ? ~0ULL : ((2ULL << (b & 63)) - 1);
            const u32 sidx = sa + s_wpre[w] + (u32)__popcll(s_mask[w] & below) - 1;   // sa > 0 when the chunk starts inside a supermer, so this never underflows
            const u32 i = j - s_koff[sidx];
            const u64 bit = sm_gpos ? (src_bit0 + 2 * (sm_gpos[seg.sup_off + first + sidx] + (u64)i))
                                    : (8 * (byte_abs + s_boff[sidx]) + 2 * (u64)i);
            Mer<NW> mer;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) mer.w[w2] = bits64_bytes_clamped(src8, bit + 64 * w2, src_words);
            mer.w[NW - 1] &= lastmask;
            Mer<NW> cm = canonical<NW>(mer, k);
            const u64 o = kbase + j;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) keys_out[o * NW + w2] = cm.w[w2];
            if (EXT) {
                const u64 sa_abs = seg.sup_off + first + sidx;
                vals_out[o] = (u64)(sm_pos[sa_abs] + i) | ((u64)(u32)sm_rid[sa_abs] << 32);
            }
        }
        __syncthreads();
    }
    (void)lane;
}

// Multi-GPU only: materialise the re-aligned byte stream of reference-mode supermers for the
// exchange.  One lane per output byte (coalesced), the supermer of a byte is found by binary search
// over the tile's byte prefix sums; layout identical to what parse_kernel's copy mode writes.
__global__ __launch_bounds__(EXP_THREADS) void pack_kernel(const ExpSeg *segs, int nseg, const u8 *sm_len, const u64 *src8, u64 src_bit0, u64 src_words,
                                                            const u64 *sm_gpos, const u64 *tile_off, u8 *bytes_out)
{
    __shared__ u32 s_boff[EXP_TILE + 1];
    __shared__ u32 s_scr[8];
    const u64 tile = blockIdx.x;
    const int sg = seg_of_tile(segs, nseg, tile);
    const ExpSeg seg = segs[sg];
    const u64 first = (tile - seg.tile_start) * EXP_TILE;
    const u32 ns = (u32)((seg.n_sup - first) < (u64)EXP_TILE ? (seg.n_sup - first) : (u64)EXP_TILE);
    const int tid = threadIdx.x;
    u32 nb[EXP_SPT], sb = 0;
#pragma unroll
    for (int i = 0; i < EXP_SPT; ++i) {
        u32 s = tid * EXP_SPT + i;
        u32 len = (s < ns) ? sm_len[seg.sup_off + first + s] : 0;
        nb[i] = (s < ns) ? ((len + 3) >> 2) : 0;
        sb += nb[i];
    }
    u32 totb;
    u32 eb = block_excl_scan_256<u32>(sb, s_scr, &totb);
#pragma unroll
    for (int i = 0; i < EXP_SPT; ++i) { s_boff[tid * EXP_SPT + i] = eb; eb += nb[i]; }
    if (tid == EXP_THREADS - 1) s_boff[EXP_TILE] = eb;
    __syncthreads();
    const u64 byte_abs = tile_off[2 * tile];
    for (u32 b = tid; b < totb; b += EXP_THREADS) {
        u32 lo = 0, hi = ns - 1;
        while (lo < hi) { u32 mid = (lo + hi + 1) >> 1; if (s_boff[mid] <= b) lo = mid; else hi = mid - 1; }
        const u32 jb = b - s_boff[lo];
        const u32 len = sm_len[seg.sup_off + first + lo];
        const u64 bit = src_bit0 + 2 * sm_gpos[seg.sup_off + first + lo] + 8 * (u64)jb;
        u8 byte = (u8)(bits64_bytes_clamped(src8, bit, src_words) >> 56);
        const u32 nbs = (len + 3) >> 2;
        if (jb == nbs - 1 && (len & 3)) byte &= (u8)(0xFF << (2 * (4 - (len & 3))));
        bytes_out[byte_abs + b] = byte;
    }
}

} // namespace hsk
