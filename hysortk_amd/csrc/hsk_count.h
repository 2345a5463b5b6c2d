// hsk_count.h -- adjacent-equal merge-count with the [L,U] frequency filter (device).
//
// Replaces count_sorted_kmers (reference src/kmerops.cpp:1410-1445): a run of equal keys in the
// sorted task array becomes one (k-mer, count) entry, kept iff L <= count <= U; with EXTENSION the
// run's (PosInRead, ReadId) payloads follow in sorted-array order (kmerops.cpp:1430-1437).
//
// The reference scans serially per task.  Here a tile of 2048 sorted records sits in LDS, every
// lane owns 8 consecutive records, a run is owned by the lane holding its first record, and that
// lane measures the run (walking LDS, then HBM if the run leaves the tile; the walk stops at U+1
// because longer runs are dropped anyway).  Two launches -- COUNT (kept runs / payloads per tile),
// exclusive scan, EMIT -- keep the output in sorted order with no atomics; the count histogram
// (print_kmer_histogram, reference src/hysortk.cpp:98-136) is accumulated per workgroup in LDS.
#pragma once
#include "hsk_device.h"

namespace hsk {

constexpr int CNT_THREADS = 256;
constexpr int CNT_PPT = 8;
constexpr int CNT_TILE = CNT_THREADS * CNT_PPT;
constexpr int CNT_LDS_HIST = 1024;       // counts below this are histogrammed in LDS

struct CountArgs {
    const u64 *keys;        // sorted, n * NW
    const u64 *vals;        // EXTENSION payload (pos | rid << 32), same order
    u64 n;
    u32 lower, upper;
    u64 *tile_cnt;          // COUNT out / EMIT in (after scan: exclusive offsets) [ntiles][2] = {entries, payloads}
    u64 *entries;           // EMIT: (NW + 1) words per entry
    u64 *payoff;            // EMIT, EXT: per entry payload start (relative to this task's payload base)
    u32 *pos; int32_t *rid; // EMIT, EXT
    u64 pay_base;           // payload offset of this task inside pos/rid
    u64 *histo;             // [histo_len]
    u32 histo_len;
};

template <int NW>
__device__ __forceinline__ bool keys_equal(const u64 *a, const u64 *b)
{
    bool e = true;
#pragma unroll
    for (int w = 0; w < NW; ++w) e = e && (a[w] == b[w]);
    return e;
}

template <int NW, bool EMIT, bool EXT>
__global__ __launch_bounds__(CNT_THREADS) void count_kernel(CountArgs a)
{
    __shared__ u64 s_k[(CNT_TILE + 1) * NW];     // [0] = record preceding the tile
    __shared__ u64 s_scr[8];
    __shared__ u32 s_hist[EMIT ? CNT_LDS_HIST : 1];
    const int tid = threadIdx.x;
    const u64 tile = blockIdx.x;
    const u64 base = tile * CNT_TILE;
    const u32 tn = (u32)((a.n - base) < (u64)CNT_TILE ? (a.n - base) : (u64)CNT_TILE);

    if (EMIT) for (int i = tid; i < CNT_LDS_HIST; i += CNT_THREADS) s_hist[i] = 0;
    // stage records base-1 .. base+tn-1 (coalesced)
    for (u32 i = tid; i < (tn + 1) * NW; i += CNT_THREADS) {
        const long long gi = (long long)(base * NW) + (long long)i - NW;
        s_k[i] = gi >= 0 ? a.keys[gi] : 0;
    }
    __syncthreads();

    u32 run_len[CNT_PPT];      // 0: not a kept run start
    u32 nkeep = 0; u64 npay = 0;
#pragma unroll
    for (int i = 0; i < CNT_PPT; ++i) {
        const u32 p = tid * CNT_PPT + i;
        run_len[i] = 0;
        if (p >= tn) continue;
        const u64 *me = &s_k[(p + 1) * NW];
        const bool start = (base + p == 0) || !keys_equal<NW>(me, &s_k[p * NW]);
        if (!start) continue;
        u32 c = 1, q = p + 1;
        while (q < tn && c <= a.upper && keys_equal<NW>(&s_k[(q + 1) * NW], me)) { ++q; ++c; }
        if (q == tn && c <= a.upper) {                 // the run may continue in the following tiles
            u64 g = base + tn;
            while (g < a.n && c <= a.upper) {
                u64 o[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) o[w] = a.keys[g * NW + w];
                if (!keys_equal<NW>(o, me)) break;
                ++g; ++c;
            }
        }
        if (c >= a.lower && c <= a.upper) { run_len[i] = c; ++nkeep; npay += c; }
    }

    if (!EMIT) {
        u64 tk, tp;
        block_excl_scan_256<u64>((u64)nkeep, s_scr, &tk);
        block_excl_scan_256<u64>(npay, s_scr, &tp);
        if (tid == 0) { a.tile_cnt[2 * tile] = tk; a.tile_cnt[2 * tile + 1] = tp; }
        return;
    } else {
        u64 tk, tp;
        u64 ek = block_excl_scan_256<u64>((u64)nkeep, s_scr, &tk);
        u64 ep = EXT ? block_excl_scan_256<u64>(npay, s_scr, &tp) : 0;
        u64 o = a.tile_cnt[2 * tile] + ek;
        u64 po = EXT ? (a.tile_cnt[2 * tile + 1] + ep) : 0;
#pragma unroll
        for (int i = 0; i < CNT_PPT; ++i) {
            const u32 c = run_len[i];
            if (!c) continue;
            const u32 p = tid * CNT_PPT + i;
#pragma unroll
            for (int w = 0; w < NW; ++w) a.entries[o * (NW + 1) + w] = s_k[(p + 1) * NW + w];
            a.entries[o * (NW + 1) + NW] = c;
            if (c < CNT_LDS_HIST) atomicAdd(&s_hist[c], 1u);
            else if (c < a.histo_len) atomicAdd((unsigned long long *)&a.histo[c], 1ULL);
            if (EXT) {
                a.payoff[o] = po;
                for (u32 j = 0; j < c; ++j) {
                    const u64 v = a.vals[base + p + j];
                    a.pos[a.pay_base + po + j] = (u32)v;
                    a.rid[a.pay_base + po + j] = (int32_t)(v >> 32);
                }
                po += c;
            }
            ++o;
        }
        __syncthreads();
        for (int i = tid; i < CNT_LDS_HIST; i += CNT_THREADS) {
            const u32 c = s_hist[i];
            if (c && (u32)i < a.histo_len) atomicAdd((unsigned long long *)&a.histo[i], (unsigned long long)c);
        }
    }
}

// single workgroup: in-place exclusive scan of tile_cnt[ntiles][2]; totals to total[2]
__global__ __launch_bounds__(CNT_THREADS) void count_scan_kernel(u64 *tile_cnt, u64 ntiles, u64 *total)
{
    __shared__ u64 s_scr[8];
    __shared__ u64 s_carry[2];
    if (threadIdx.x == 0) { s_carry[0] = 0; s_carry[1] = 0; }
    __syncthreads();
    for (u64 b = 0; b < ntiles; b += CNT_THREADS) {
        const u64 t = b + threadIdx.x;
        u64 k = 0, p = 0;
        if (t < ntiles) { k = tile_cnt[2 * t]; p = tile_cnt[2 * t + 1]; }
        u64 tk, tp;
        u64 ek = block_excl_scan_256<u64>(k, s_scr, &tk);
        u64 ep = block_excl_scan_256<u64>(p, s_scr, &tp);
        if (t < ntiles) { tile_cnt[2 * t] = s_carry[0] + ek; tile_cnt[2 * t + 1] = s_carry[1] + ep; }
        __syncthreads();
        if (threadIdx.x == 0) { s_carry[0] += tk; s_carry[1] += tp; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { total[0] = s_carry[0]; total[1] = s_carry[1]; }
}

} // namespace hsk
