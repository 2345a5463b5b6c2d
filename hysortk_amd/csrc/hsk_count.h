// hsk_count.h -- adjacent-equal merge-count with the [L,U] frequency filter (device).
//
// Replaces count_sorted_kmers (reference src/kmerops.cpp:1410-1445): a run of equal keys in the
// sorted task array becomes one (k-mer, count) entry, kept iff L <= count <= U; with EXTENSION the
// run's (PosInRead, ReadId) payloads are the run's slice of the sorted payload array
// (kmerops.cpp:1430-1437 copies exactly that slice, in sorted-array order).
//
// The reference scans serially per task.  Here a tile of 2048 sorted records is staged in LDS with
// coalesced loads; lane t looks at records t, t+256, ... (conflict-free LDS reads of neighbours);
// run heads become a 2048-bit mask built with wave ballots; a head finds the end of its run with
// find-first-set over that mask; only the last run of a tile may have to look into HBM (bounded by
// U+1, longer runs are dropped anyway).  Output slots are popcount prefixes of the "kept" mask, so
// the list stays in sorted order with no atomics.  Two launches (COUNT per tile, scan, EMIT); the
// count histogram (print_kmer_histogram, reference src/hysortk.cpp:98-136) is accumulated per
// workgroup in LDS.
#pragma once
#include "hsk_device.h"

namespace hsk {

constexpr int CNT_THREADS = 256;
constexpr int CNT_PPT = 8;
constexpr int CNT_TILE = CNT_THREADS * CNT_PPT;
constexpr int CNT_WORDS = CNT_TILE / 64;     // 32 mask words
constexpr int CNT_LDS_HIST = 256;            // counts below this are histogrammed in LDS
constexpr int CNT_HALO = 256;                // records of the next tile staged too: the tile's last run usually ends there

struct CountArgs {
    const u64 *keys;        // sorted, n * NW
    u64 n;
    u32 lower, upper;
    u64 *tile_cnt;          // COUNT out / EMIT in (after the scan: exclusive entry offsets) [ntiles]
    u64 *entries;           // EMIT: (NW + 1) words per entry
    u64 *run_start;         // EMIT, EXT: first record of the run in the sorted array (+ payoff_add)
    u64 payoff_add;
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
    __shared__ u64 s_k[(CNT_TILE + 1 + CNT_HALO) * NW];     // [0] = record preceding the tile, then the tile, then the halo
    __shared__ u64 s_head[CNT_WORDS];
    __shared__ u64 s_keep[CNT_WORDS];
    __shared__ u32 s_pre[CNT_WORDS + 1];
    __shared__ u32 s_hist[EMIT ? CNT_LDS_HIST : 1];
    __shared__ u16 s_ocnt[EMIT ? CNT_TILE : 2];     // count of the e-th kept run of the tile (U <= 65535)
    __shared__ u16 s_opos[EMIT ? CNT_TILE : 2];     // its first record
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 ntiles = (a.n + CNT_TILE - 1) / CNT_TILE;
    if (EMIT) for (int i = tid; i < CNT_LDS_HIST; i += CNT_THREADS) s_hist[i] = 0;
    // workgroups stride over the tiles: the LDS count histogram is flushed once per workgroup, not per tile
    // (a per-tile flush is ~30 global atomics on the same few hot bins per 2048 records)
    for (u64 tile = blockIdx.x; tile < ntiles; tile += (EMIT ? (u64)gridDim.x : ntiles)) {     // COUNT: one tile per workgroup
    const u64 base = tile * CNT_TILE;
    const u32 tn = (u32)((a.n - base) < (u64)CNT_TILE ? (a.n - base) : (u64)CNT_TILE);
    if (EMIT) __syncthreads();
    const u64 obase = EMIT ? a.tile_cnt[tile] : 0;             // issued early: not on the critical path at the end
    const u32 hn = (u32)((a.n - base - tn) < (u64)CNT_HALO ? (a.n - base - tn) : (u64)CNT_HALO);   // halo records available
    for (u32 i = tid; i < (tn + 1 + hn) * NW; i += CNT_THREADS) {     // records base-1 .. base+tn+hn-1
        const long long gi = (long long)(base * NW) + (long long)i - NW;
        s_k[i] = gi >= 0 ? a.keys[gi] : 0;
    }
    __syncthreads();

    // ---- run heads -> bit mask --------------------------------------------------------------------
    bool head[CNT_PPT];
#pragma unroll
    for (int j = 0; j < CNT_PPT; ++j) {
        const u32 p = j * CNT_THREADS + tid;
        head[j] = p < tn && ((base + p == 0) || !keys_equal<NW>(&s_k[(p + 1) * NW], &s_k[p * NW]));
        const u64 m = __ballot(head[j]);
        if (lane == 0) s_head[j * 4 + wave] = m;
    }
    __syncthreads();

    // ---- run lengths, filter ------------------------------------------------------------------------
    u32 runlen[CNT_PPT];
#pragma unroll
    for (int j = 0; j < CNT_PPT; ++j) {
        const u32 p = j * CNT_THREADS + tid;
        u32 c = 0;
        if (head[j]) {
            u32 w = p >> 6;
            u64 m = s_head[w] & ((p & 63) == 63 ? 0ULL : (~0ULL << ((p & 63) + 1)));
            while (m == 0 && ++w < (u32)CNT_WORDS) m = s_head[w];
            if (m) c = (w << 6) + (u32)__builtin_ctzll(m) - p;
            else {                                            // last run of the tile: may continue in HBM
                c = tn - p;
                const u64 *me = &s_k[(p + 1) * NW];
                u32 h = 0;                                    // first the halo in LDS (a dependent HBM load per step is ~1 us)
                while (h < hn && c <= a.upper && keys_equal<NW>(&s_k[(tn + 1 + h) * NW], me)) { ++h; ++c; }
                u64 g = base + tn + h;
                while (h == hn && g < a.n && c <= a.upper) {
                    u64 o[NW];
#pragma unroll
                    for (int x = 0; x < NW; ++x) o[x] = a.keys[g * NW + x];
                    if (!keys_equal<NW>(o, me)) break;
                    ++g; ++c;
                }
            }
        }
        const bool keep = c >= a.lower && c <= a.upper;       // c == 0 for non-heads (lower >= 1)
        runlen[j] = keep ? c : 0;
        const u64 km = __ballot(keep);
        if (lane == 0) s_keep[j * 4 + wave] = km;
    }
    __syncthreads();
    if (tid < 64) {                                           // exclusive popcount prefix over the 32 mask words
        u32 v = tid < CNT_WORDS ? (u32)__popcll(s_keep[tid]) : 0;
        u32 inc = wave_incl_scan<u32>(v);
        if (tid < CNT_WORDS) s_pre[tid] = inc - v;
        if (tid == CNT_WORDS - 1) s_pre[CNT_WORDS] = inc;
    }
    __syncthreads();

    if (!EMIT) {
        if (tid == 0) a.tile_cnt[tile] = s_pre[CNT_WORDS];
    } else {
        // kept runs are first listed in LDS in output order (slot = popcount prefix), then the entries
        // are written by consecutive lanes to consecutive records: coalesced 16/24/32-byte records
        // instead of one scattered record per lane
#pragma unroll
        for (int j = 0; j < CNT_PPT; ++j) {
            const u32 c = runlen[j];
            if (!c) continue;
            const u32 p = j * CNT_THREADS + tid;
            const u32 w = j * 4 + wave;
            const u32 slot = s_pre[w] + (u32)__popcll(s_keep[w] & ((1ULL << lane) - 1));
            s_opos[slot] = (u16)p; s_ocnt[slot] = (u16)c;
            if (c < CNT_LDS_HIST) atomicAdd(&s_hist[c], 1u);
            else if (c < a.histo_len) atomicAdd((unsigned long long *)&a.histo[c], 1ULL);
        }
        __syncthreads();
        const u32 nkept = s_pre[CNT_WORDS];
        for (u32 i = tid; i < nkept * (NW + 1); i += CNT_THREADS) {      // one 8-byte word per lane, fully coalesced
            const u32 e = i / (NW + 1), x = i - e * (NW + 1);
            const u32 p = s_opos[e];
            a.entries[(obase + e) * (NW + 1) + x] = (x < (u32)NW) ? s_k[(p + 1) * NW + x] : (u64)s_ocnt[e];
        }
        if (EXT) for (u32 e = tid; e < nkept; e += CNT_THREADS) a.run_start[obase + e] = a.payoff_add + base + s_opos[e];
    }
    }   // tile loop
    if (EMIT) {
        __syncthreads();
        for (int i = tid; i < CNT_LDS_HIST; i += CNT_THREADS) {
            const u32 c = s_hist[i];
            if (c && (u32)i < a.histo_len) atomicAdd((unsigned long long *)&a.histo[i], (unsigned long long)c);
        }
    }
}

// single workgroup: in-place exclusive scan of tile_cnt[ntiles]; total to total[0]
constexpr int CSCAN_IPT = 8;
__global__ __launch_bounds__(CNT_THREADS) void count_scan_kernel(u64 *tile_cnt, u64 ntiles, u64 *total)
{
    __shared__ u64 s_scr[8];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 b = 0; b < ntiles; b += (u64)CNT_THREADS * CSCAN_IPT) {
        const u64 t0 = b + (u64)threadIdx.x * CSCAN_IPT;
        u64 v[CSCAN_IPT], s = 0;
#pragma unroll
        for (int i = 0; i < CSCAN_IPT; ++i) { v[i] = (t0 + i < ntiles) ? tile_cnt[t0 + i] : 0; s += v[i]; }
        u64 tot;
        u64 e = block_excl_scan_256<u64>(s, s_scr, &tot) + s_carry;
#pragma unroll
        for (int i = 0; i < CSCAN_IPT; ++i) { if (t0 + i < ntiles) tile_cnt[t0 + i] = e; e += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = s_carry;
}

// EXTENSION: split the sorted payload words (pos | rid << 32) into the two output arrays
__global__ void payload_split_kernel(const u64 *vals, u64 n, u32 *pos, int32_t *rid)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 v = vals[i];
        pos[i] = (u32)v; rid[i] = (int32_t)(v >> 32);
    }
}

// Result egress over PCIe (hsk_count() to host memory): an entry {k-mer words, u64 count} whose count fits 16 bits (the config
// contract: UPPER_KMER_FREQ <= 65535, reference include/compiletime.h:21) travels as k-mer words + u16 -- 10 instead of 16 bytes
// for one-word keys -- and is widened to the KmerListEntryS layout (reference include/kmer.hpp:368) by host threads.
// keys_out: n * nw words, cnt_out: n values; one launch per task.
__global__ __launch_bounds__(256) void pack_entries_kernel(const u64 *entries, u64 n, int nw, u64 *keys_out, unsigned short *cnt_out)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 *e = entries + i * (u64)(nw + 1);
        for (int w = 0; w < nw; ++w) keys_out[i * (u64)nw + w] = e[w];
        cnt_out[i] = (unsigned short)e[nw];
    }
}

// One-word keys, the smaller form: a task's list is ascending, so the top 16 key bits -- the prefix the sort grouped by -- need
// not travel with every entry: an entry is the low 48 key bits (u32 + u16) and its count (u8 when UPPER_KMER_FREQ <= 255, else
// u16), 7 bytes for 16, and a directory per task says where each prefix starts: dir[p] = first entry whose key >> 48 is >= p,
// dir[65536] = n (first written where a prefix begins, then closed over the empty prefixes by pack_dir_close_kernel).
template <typename CT>
__global__ __launch_bounds__(256) void pack_entries_prefix_kernel(const u64 *entries, u64 n, u32 *lo32, unsigned short *mid16, CT *cnt, u32 *dir)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 key = entries[2 * i];
        lo32[i] = (u32)key; mid16[i] = (unsigned short)(key >> 32); cnt[i] = (CT)entries[2 * i + 1];
        const u32 p = (u32)(key >> 48);
        if (i == 0 || (u32)(entries[2 * i - 2] >> 48) != p) dir[p] = (u32)i;
    }
}
// dir[p] = 0xFFFFFFFF for prefixes without entries -> the start of the next prefix that has some (suffix minimum); one workgroup
// of 1024 threads per task, 64 prefixes per thread; dir[65536] = n.
__global__ __launch_bounds__(1024) void pack_dir_close_kernel(u32 *dir, u32 n)
{
    __shared__ u32 s_min[1024];
    const int tid = threadIdx.x;
    u32 m = 0xFFFFFFFFu;
    for (int q = 63; q >= 0; --q) { const u32 v = dir[tid * 64 + q]; if (v < m) m = v; }
    s_min[tid] = m;
    __syncthreads();
    // suffix minimum over the threads behind this one (1024 values: a plain loop per thread would do; halving steps are shorter)
    for (int d = 1; d < 1024; d <<= 1) {
        const u32 o = (tid + d < 1024) ? s_min[tid + d] : 0xFFFFFFFFu;
        __syncthreads();
        if (o < s_min[tid]) s_min[tid] = o;
        __syncthreads();
    }
    u32 run = (tid + 1 < 1024) ? s_min[tid + 1] : 0xFFFFFFFFu;
    if (run > n) run = n;
    for (int q = 63; q >= 0; --q) { const u32 v = dir[tid * 64 + q]; if (v < run) run = v; dir[tid * 64 + q] = run; }
    if (tid == 0) dir[65536] = n;
}

} // namespace hsk
