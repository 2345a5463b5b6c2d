// hsk_heavy.h -- the receiving side of heavy-hitter pre-aggregation (device).
//
// Replaces GatheredKmerList::process + count_sorted_kmerlist (reference src/kmerops.cpp:546-581, 1447-1480).
// A task whose global k-mer count exceeds UNBALANCED_RATIO x the mean (HeavyHitterClassifier, kmerops.cpp:1157)
// does not travel as supermers: every rank extracts, sorts and counts its OWN k-mers of the task without the
// frequency filter (ScatteredKmerList, kmerops.cpp:363-398; here: the ordinary expand / sort / aggregate kernels
// with L = 1, U = max) and ships the (k-mer, count) list; the owner concatenates the lists, orders the entries
// by key (hsk_sort.h with the count as payload), SUMS the counts of equal keys and applies [L, U].
// Every source list holds a key at most once, so a run of equal keys is at most `nranks` long.
#pragma once
#include "hsk_device.h"

namespace hsk {

constexpr int HV_THREADS = 256;

// entries {key words, count} -> keys[] (NW words each), counts[]
template <int NW>
__global__ __launch_bounds__(HV_THREADS) void heavy_split_kernel(const u64 *entries, u64 n, u64 *keys, u64 *cnts)
{
    const u64 stride = (u64)gridDim.x * HV_THREADS;
    for (u64 i = (u64)blockIdx.x * HV_THREADS + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int x = 0; x < NW; ++x) keys[i * NW + x] = entries[(NW + 1) * i + x];
        cnts[i] = entries[(NW + 1) * i + NW];
    }
}

struct HeavyMergeArgs {
    const u64 *keys, *cnts; u64 n;      // sorted by key (records of NW words)
    u64 lower, upper;
    u64 *tile_cnt;                      // COUNT out / EMIT in (exclusive offsets)
    u64 *entries;                       // EMIT: {key words, summed count}
    u64 *histo; u32 histo_len;
};

template <int NW>
__device__ __forceinline__ bool heavy_same(const u64 *keys, u64 i, const u64 (&k)[NW])
{
    bool eq = true;
#pragma unroll
    for (int x = 0; x < NW; ++x) eq = eq && keys[i * NW + x] == k[x];
    return eq;
}

// one record per lane and tile of 256 records: a run head sums its run (<= nranks records), kept heads are
// compacted with a block scan: the output stays in key order
template <int NW, bool EMIT>
__global__ __launch_bounds__(HV_THREADS) void heavy_merge_kernel(HeavyMergeArgs a)
{
    __shared__ u32 s_scr[8];
    const u64 i = (u64)blockIdx.x * HV_THREADS + threadIdx.x;
    u64 key[NW], sum = 0; bool keep = false;
#pragma unroll
    for (int x = 0; x < NW; ++x) key[x] = 0;
    if (i < a.n) {
#pragma unroll
        for (int x = 0; x < NW; ++x) key[x] = a.keys[i * NW + x];
        const bool head = (i == 0) || !heavy_same<NW>(a.keys, i - 1, key);
        if (head) {
            sum = a.cnts[i];
            for (u64 j = i + 1; j < a.n && heavy_same<NW>(a.keys, j, key); ++j) sum += a.cnts[j];
            keep = sum >= a.lower && sum <= a.upper;
        }
    }
    u32 tot;
    const u32 o = block_excl_scan_256<u32>(keep ? 1u : 0u, s_scr, &tot);
    if (!EMIT) { if (threadIdx.x == 0) a.tile_cnt[blockIdx.x] = tot; return; }
    if (keep) {
        const u64 e = a.tile_cnt[blockIdx.x] + o;
#pragma unroll
        for (int x = 0; x < NW; ++x) a.entries[(NW + 1) * e + x] = key[x];
        a.entries[(NW + 1) * e + NW] = sum;
        if (sum < a.histo_len) atomicAdd((unsigned long long *)&a.histo[sum], 1ULL);
    }
}

} // namespace hsk
