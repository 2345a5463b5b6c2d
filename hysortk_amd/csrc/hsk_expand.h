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
constexpr int EXP_SPT = 4;
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
// NW + 1 words of the base stream starting at bit `bit` (bits past the buffer read as zero)
template <int NW>
__device__ __forceinline__ void load_window(const u64 *p8, u64 bit, u64 nwords, u64 (&w)[NW + 1])
{
    const u64 i = bit >> 6; const u32 s = (u32)(bit & 63);
    u64 a[NW + 2];
#pragma unroll
    for (int x = 0; x < NW + 2; ++x) a[x] = (i + x < nwords) ? __builtin_bswap64(p8[i + x]) : 0;
#pragma unroll
    for (int x = 0; x < NW + 1; ++x) w[x] = s ? ((a[x] << s) | (a[x + 1] >> (64 - s))) : a[x];
}

// The k-mers of a tile are produced by WORK ITEMS: a supermer of n k-mers is cut into ceil(n / 8) items of up to
// 8 consecutive k-mers.  A lane takes one item: it pulls a window of the base stream once (all loads of a lane are
// issued together, nothing waits inside the roll) and then ROLLS: the forward k-mer is the window shifted by one
// base per step, its twin is shifted the other way with the complement of the entering base on top
// (the reference recomputes Kmer::GetTwin per k-mer, include/kmer.hpp:266-296).  The 256 items of one step cover a
// contiguous range of at most 2048 output records; they pass through an LDS buffer so that the global stores are
// coalesced (consecutive lanes, consecutive records).
constexpr int EXP_RUN = 8;                            // k-mers per work item
constexpr int EXP_OUT = EXP_THREADS * EXP_RUN;        // records per step at most
constexpr int EXP_OUT_LDS = EXP_OUT + EXP_OUT / 8;    // one pad record per 8 (items start 8 records apart: spreads the banks)

template <int NW, bool EXT>
__global__ __launch_bounds__(EXP_THREADS) void expand_kernel(const ExpSeg *segs, int nseg, const u8 *sm_len, const u64 *src8, u64 src_bit0, u64 src_words,
                                                              const u64 *sm_gpos, const u32 *sm_pos, const int32_t *sm_rid, const u64 *tile_off,
                                                              int k, u64 *keys_out, u64 *vals_out)
{
    __shared__ u32 s_boff[EXP_TILE + 1];
    __shared__ u32 s_koff[EXP_TILE + 1];
    __shared__ u32 s_ioff[EXP_TILE + 1];
    __shared__ u32 s_scr[8];
    __shared__ u32 s_rng[2];
    __shared__ u64 s_out[EXP_OUT_LDS];
    const u64 tile = blockIdx.x;
    const int sg = seg_of_tile(segs, nseg, tile);
    const ExpSeg seg = segs[sg];
    const u64 first = (tile - seg.tile_start) * EXP_TILE;
    const u32 ns = (u32)((seg.n_sup - first) < (u64)EXP_TILE ? (seg.n_sup - first) : (u64)EXP_TILE);
    const int tid = threadIdx.x;

    // blocked arrangement: thread t owns supermers [t*SPT, t*SPT+SPT) of the tile
    u32 nb[EXP_SPT], nk[EXP_SPT], sb = 0, sk = 0, si = 0;
#pragma unroll
    for (int i = 0; i < EXP_SPT; ++i) {
        u32 s = tid * EXP_SPT + i;
        u32 len = (s < ns) ? sm_len[seg.sup_off + first + s] : 0;
        nb[i] = (s < ns) ? ((len + 3) >> 2) : 0;
        nk[i] = (s < ns) ? (len - k + 1) : 0;
        sb += nb[i]; sk += nk[i]; si += (nk[i] + EXP_RUN - 1) / EXP_RUN;
    }
    u32 totb, totk, toti;
    u32 eb = block_excl_scan_256<u32>(sb, s_scr, &totb);
    u32 ek = block_excl_scan_256<u32>(sk, s_scr, &totk);
    u32 ei = block_excl_scan_256<u32>(si, s_scr, &toti);
#pragma unroll
    for (int i = 0; i < EXP_SPT; ++i) {
        s_boff[tid * EXP_SPT + i] = eb; s_koff[tid * EXP_SPT + i] = ek; s_ioff[tid * EXP_SPT + i] = ei;
        eb += nb[i]; ek += nk[i]; ei += (nk[i] + EXP_RUN - 1) / EXP_RUN;
    }
    if (tid == EXP_THREADS - 1) { s_boff[EXP_TILE] = eb; s_koff[EXP_TILE] = ek; s_ioff[EXP_TILE] = ei; }
    __syncthreads();

    const u64 byte_abs = tile_off[2 * tile];
    const u64 kbase = tile_off[2 * tile + 1];
    const int low = 64 * NW - 2 * k;                      // unused low bits of the last word; 0 < low < 64 (k % 32 != 0)
    const u64 lastmask = ~0ULL << low;

    for (u32 it0 = 0; it0 < toti; it0 += EXP_THREADS) {
        const u32 item = it0 + tid;
        const bool valid = item < toti;
        u32 cnt = 0, out0 = 0;
        u64 keys[EXP_RUN][NW];
        u64 vals[EXT ? EXP_RUN : 1];
        if (valid) {
            u32 lo = 0, hi = ns;                           // last supermer whose first item is <= item (every supermer has >= 1 item)
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_ioff[mid] <= item) lo = mid; else hi = mid; }
            const u32 sidx = lo;
            const u32 i0 = (item - s_ioff[sidx]) * EXP_RUN;
            const u32 k0 = s_koff[sidx];
            const u32 nks = s_koff[sidx + 1] - k0;
            cnt = nks - i0 < (u32)EXP_RUN ? nks - i0 : (u32)EXP_RUN;
            out0 = k0 + i0;
            const u64 sabs = seg.sup_off + first + sidx;
            const u64 bit = sm_gpos ? (src_bit0 + 2 * (sm_gpos[sabs] + (u64)i0)) : (8 * (byte_abs + s_boff[sidx]) + 2 * (u64)i0);
            u64 win[NW + 1];
            load_window<NW>(src8, bit, src_words, win);
            u64 vbase = 0;
            if (EXT) vbase = (u64)(sm_pos[sabs] + i0) | ((u64)(u32)sm_rid[sabs] << 32);
            Mer<NW> fw, rc;
#pragma unroll
            for (int x = 0; x < NW; ++x) fw.w[x] = win[x];
            fw.w[NW - 1] &= lastmask;
            rc = twin<NW>(fw, k);
#pragma unroll
            for (int r = 0; r < EXP_RUN; ++r) {
                if (r > 0) {
                    // one base further: window left by 2 bits; twin right by 2 bits, complement of the entering base on top
#pragma unroll
                    for (int x = 0; x < NW; ++x) win[x] = (win[x] << 2) | (win[x + 1] >> 62);
                    win[NW] <<= 2;
#pragma unroll
                    for (int x = 0; x < NW; ++x) fw.w[x] = win[x];
                    fw.w[NW - 1] &= lastmask;
                    const u64 nbase = (fw.w[NW - 1] >> low) & 3;
#pragma unroll
                    for (int x = NW - 1; x > 0; --x) rc.w[x] = (rc.w[x] >> 2) | (rc.w[x - 1] << 62);
                    rc.w[0] = (rc.w[0] >> 2) | ((3 - nbase) << 62);
                    rc.w[NW - 1] &= lastmask;
                }
                const bool use_rc = mer_less<NW>(rc, fw);
#pragma unroll
                for (int x = 0; x < NW; ++x) keys[r][x] = use_rc ? rc.w[x] : fw.w[x];
                if (EXT) vals[r] = vbase + (u64)r;         // pos + r (the low word never carries: pos < read length)
            }
        }
        if (tid == 0) s_rng[0] = out0;
        if (valid && (tid == EXP_THREADS - 1 || item + 1 == toti)) s_rng[1] = out0 + cnt;
        __syncthreads();
        const u32 o0 = s_rng[0], nout = s_rng[1] - o0;
        const u32 q0 = out0 - o0;                          // first record of this lane's item inside the step's range
        // one key word (or the payload) at a time through the LDS buffer
#pragma unroll
        for (int x = 0; x < NW + (EXT ? 1 : 0); ++x) {
            if (x > 0) __syncthreads();
#pragma unroll
            for (int r = 0; r < EXP_RUN; ++r)
                if ((u32)r < cnt) { const u32 q = q0 + r; s_out[q + (q >> 3)] = (x < NW) ? keys[r][x < NW ? x : 0] : vals[EXT ? r : 0]; }
            __syncthreads();
            for (u32 q = tid; q < nout; q += EXP_THREADS) {
                const u64 v = s_out[q + (q >> 3)];
                if (x < NW) keys_out[(kbase + o0 + q) * NW + x] = v; else vals_out[kbase + o0 + q] = v;
            }
        }
        __syncthreads();
    }
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
