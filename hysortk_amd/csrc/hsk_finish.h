// hsk_finish.h -- fused finish of the hybrid sort for one-word keys: orders the low bits inside every
// prefix bin, merge-counts and filters in ONE pass over the keys.
//
// Replaces, after the LSD passes over the top `64 - hi_shift` bits (hsk_sort.h):
//   the remaining radix passes of sort_task      reference src/kmerops.cpp:1382 (RADULS / PARADIS)
//   count_sorted_kmers                            reference src/kmerops.cpp:1410-1445
// and supersedes binsort_kernel + count_kernel<COUNT> + count_kernel<EMIT> for this case
// (12 B read + 8 B written + 2 x 8 B read per key  ->  10 B read per key + the entries).
//
// finish_kernel: a tile (2048 records + 512 of halo) is staged in LDS.  A bin (records sharing the sorted prefix)
// belongs to the tile in which it STARTS; the owner sees all of it if it ends inside the halo.  Bins holding one
// distinct key need no work (their records are one run); a bin with several keys is ordered in LDS by rank
// counting.  Runs -> counts -> [L,U] filter -> popcount compaction as in hsk_count.h.  The kept entries of a tile
// go to the tile's fixed slot range of a scratch buffer (a tile can keep at most (2048+512)/L entries) together
// with their number: no inter-workgroup dependency at all.  count_scan_kernel turns the numbers into offsets and
// finish_compact_kernel moves the entries (only those: ~0.6 B per key at L=15) to their final place and builds
// the count histogram on the way.
// What an owner cannot see (a bin running past its halo) is handled conservatively: a single-key giant bin is still
// one run (its length is walked in HBM up to U+1); any sign of two keys in a bin that its owner cannot see
// completely raises FN_FLAG_MIXED_GIANT and the host redoes that task with the full-width passes.
#pragma once
#include "hsk_device.h"
#include "hsk_sort.h"

namespace hsk {

constexpr int FN_THREADS = 256;
constexpr int FN_TILE = 2048;
constexpr int FN_HALO = 512;
constexpr int FN_NL = FN_TILE + FN_HALO;
constexpr int FN_WORDS = FN_NL / 64;              // 40
constexpr int FN_PPT = FN_NL / FN_THREADS;        // 10
constexpr int FN_LDS_HIST = 256;
constexpr u32 FN_MAX_WALK = 4096;                 // longest HBM walk for a giant single-key run (else: fallback)

enum { FN_FLAG_MIXED_GIANT = 1, FN_FLAG_LONG_WALK = 4 };

struct FinishArgs {
    const u64 *keys; u64 n;
    u64 *scratch; u32 cap_t;       // tile t writes its entries {key, count} to scratch[(t * cap_t + e) * 2 ..]
    u64 *tile_cnt;                 // [ntiles] kept entries of each tile
    u32 *flags;                    // out: FN_FLAG_*
    u32 lower, upper; int hi_shift;
};

// bits of mask word `w` that lie in positions [lo, hi)
__device__ __forceinline__ u64 word_range(int w, u32 lo, u32 hi)
{
    const u32 b0 = (u32)w * 64, b1 = b0 + 64;
    if (hi <= b0 || lo >= b1 || lo >= hi) return 0;
    u64 m = ~0ULL;
    if (lo > b0) m &= ~0ULL << (lo - b0);
    if (hi < b1) m &= (1ULL << (hi - b0)) - 1;
    return m;
}

// first set bit at or after position `from` in a mask of `nwords` words, or `none`
__device__ __forceinline__ u32 mask_next(const u64 *m, u32 from, u32 nwords, u32 none)
{
    u32 w = from >> 6;
    if (w >= nwords) return none;
    u64 x = m[w] & (~0ULL << (from & 63));
    while (x == 0 && ++w < nwords) x = m[w];
    return x ? (w * 64 + (u32)__builtin_ctzll(x)) : none;
}
// last set bit at or before position `upto`, or -1
__device__ __forceinline__ int mask_prev(const u64 *m, u32 upto)
{
    int w = (int)(upto >> 6);
    u64 x = m[w] & (((upto & 63) == 63) ? ~0ULL : ((2ULL << (upto & 63)) - 1));
    while (x == 0 && w > 0) x = m[--w];
    return x ? (w * 64 + 63 - (int)__builtin_clzll(x)) : -1;
}
// any set bit of (a & ~b) in positions [lo, hi)
__device__ __forceinline__ bool mask_any_andnot(const u64 *a, const u64 *b, u32 lo, u32 hi)
{
    if (lo >= hi) return false;
    const u32 w0 = lo >> 6, w1 = (hi - 1) >> 6;
    for (u32 w = w0; w <= w1; ++w) {
        u64 x = a[w] & ~b[w];
        if (w == w0) x &= ~0ULL << (lo & 63);
        if (w == w1 && (hi & 63)) x &= (1ULL << (hi & 63)) - 1;
        if (x) return true;
    }
    return false;
}


__global__ __launch_bounds__(FN_THREADS) void finish_kernel(FinishArgs a)
{
    __shared__ u64 s_k[FN_NL + 1];              // [0] = record before the tile
    __shared__ u64 s_head[FN_WORDS], s_diff[FN_WORDS], s_run[FN_WORDS], s_keep[FN_WORDS];
    __shared__ u32 s_pre[FN_WORDS + 1];
    __shared__ u16 s_opos[FN_NL], s_ocnt[FN_NL];
    __shared__ long long s_x;
    __shared__ u32 s_reg[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 tile = blockIdx.x;
    const u64 base = tile * FN_TILE;
    const u32 tn = (u32)((a.n - base) < (u64)FN_TILE ? (a.n - base) : (u64)FN_TILE);
    u64 hi = base + FN_TILE + FN_HALO; if (hi > a.n) hi = a.n;
    const u32 nl = (u32)(hi - base);
    const bool at_end = (hi == a.n);

    for (u32 i = tid; i < nl + 1; i += FN_THREADS) {
        const long long g = (long long)base + (long long)i - 1;
        s_k[i] = g >= 0 ? a.keys[g] : 0;
    }
    __syncthreads();
    for (u32 q0 = wave * 64; q0 < (u32)FN_NL; q0 += FN_THREADS) {
        const u32 q = q0 + lane;
        bool h = false, d = false;
        if (q < nl) {
            const u64 k = s_k[q + 1], kp = s_k[q];
            const bool first = (base + q == 0);
            h = first || ((k >> a.hi_shift) != (kp >> a.hi_shift));
            d = first || (k != kp);
        }
        const u64 mh = __ballot(h), md = __ballot(d);
        if (lane == 0) { s_head[q0 >> 6] = mh; s_diff[q0 >> 6] = md; }
    }
    __syncthreads();

    // ---- what this tile owns: bins starting in [0, tn).  Computed ONCE by wave 0, one mask word per lane
    //      (every lane scanning the masks itself costs more than the rest of the kernel). -------------------------
    if (wave == 0) {
        const u64 hw = lane < FN_WORDS ? s_head[lane] : 0;
        const u64 nh = lane < FN_WORDS ? (s_diff[lane] & ~hw) : 0;        // records that differ from their predecessor inside a bin
        // first head in [0, tn)
        const u64 b0 = __ballot((hw & word_range(lane, 0, tn)) != 0);
        u32 rs = tn;
        if (b0) { const int l = __builtin_ctzll(b0); const u64 x = __shfl(hw & word_range(lane, 0, tn), l, WAVE); rs = (u32)l * 64 + (u32)__builtin_ctzll(x); }
        // first head in [tn, nl)
        const u64 b1 = __ballot((hw & word_range(lane, tn, nl)) != 0);
        u32 re = nl;
        if (b1) { const int l = __builtin_ctzll(b1); const u64 x = __shfl(hw & word_range(lane, tn, nl), l, WAVE); re = (u32)l * 64 + (u32)__builtin_ctzll(x); }
        if (rs >= tn) re = rs;                                           // no bin starts in this tile: it owns nothing
        const bool tail_giant = (rs < tn) && (re == nl) && !at_end;      // the last owned bin runs past the halo
        u32 bs_last = 0;
        if (tail_giant) {                                                // its start = last head of the staged range
            const u64 b2 = __ballot((hw & word_range(lane, 0, nl)) != 0);
            if (b2) { const int l = 63 - __builtin_clzll(b2); const u64 x = __shfl(hw & word_range(lane, 0, nl), l, WAVE); bs_last = (u32)l * 64 + 63 - (u32)__builtin_clzll(x); }
        }
        const u32 re_n = tail_giant ? bs_last : re;
        const u32 lead = rs < nl ? rs : nl;
        const bool lead_mixed = __ballot((nh & word_range(lane, 0, lead)) != 0) != 0;
        const bool tail_mixed = tail_giant && __ballot((nh & word_range(lane, bs_last + 1, nl)) != 0) != 0;
        const bool mixed = __ballot((nh & word_range(lane, rs, re_n)) != 0) != 0;
        if (lane == 0) {
            s_reg[0] = rs; s_reg[1] = re; s_reg[2] = re_n; s_reg[3] = bs_last;
            s_reg[4] = (tail_giant ? 1u : 0u) | (lead_mixed ? 2u : 0u) | (tail_mixed ? 4u : 0u) | (mixed ? 8u : 0u);
        }
    }
    __syncthreads();
    const u32 rs = s_reg[0], re = s_reg[1], re_n = s_reg[2], bs_last = s_reg[3];
    const bool tail_giant = s_reg[4] & 1u, lead_mixed = s_reg[4] & 2u, tail_mixed = s_reg[4] & 4u, mixed = s_reg[4] & 8u;
    (void)bs_last;

    // ---- records before the first owned bin belong to a bin that started earlier.  If they show a second key,
    //      the owner must be able to see that bin completely, otherwise the task needs the long way. ----------
    const u32 lead = rs < nl ? rs : nl;
    if (lead_mixed) {                                                    // rare path
        if (wave == 0) {
            // wave-parallel look back (at most one tile) for the start of that bin
            const u64 pfx = s_k[1] >> a.hi_shift;
            const u64 back = base < (u64)FN_TILE ? base : (u64)FN_TILE;
            long long x = -1;                                            // bin start (global) if it lies in the previous tile
            for (u64 done = 0; done < back && x < 0; done += 64) {
                const long long g = (long long)base - 1 - (long long)done - lane;
                const bool inr = g >= (long long)(base - back);
                const bool differs = inr && ((a.keys[g] >> a.hi_shift) != pfx);
                const u64 m = __ballot(differs);
                if (m) x = (long long)base - (long long)done - (long long)__builtin_ctzll(m);   // first differing record is at base-1-done-l: the bin starts right after it
            }
            if (x < 0 && back == base) x = 0;                            // reached the start of the array inside the bin
            if (lane == 0) s_x = x;
        }
        __syncthreads();
        if (tid == 0) {
            const long long x = s_x;
            bool owner_sees_all = false;
            if (x >= 0 && (u64)x < base) {
                const u64 owner_base = ((u64)x / FN_TILE) * FN_TILE;
                u64 vis_end = owner_base + FN_TILE + FN_HALO; if (vis_end > a.n) vis_end = a.n;
                const u64 e = base + lead;
                owner_sees_all = (rs < nl || at_end) && e <= vis_end;
            }
            if (!owner_sees_all) atomicOr(a.flags, (u32)FN_FLAG_MIXED_GIANT);
        }
    }
    if (tail_mixed) { if (tid == 0) atomicOr(a.flags, (u32)FN_FLAG_MIXED_GIANT); }

    // ---- order the records of multi-key bins in place (rank counting; keys travel through registers) ---------
    if (mixed) {                                         // uniform over the workgroup
        u64 ke[FN_PPT]; u32 dst[FN_PPT];
#pragma unroll
        for (int j = 0; j < FN_PPT; ++j) {
            const u32 q = j * FN_THREADS + tid;
            dst[j] = 0xFFFFFFFFu;
            if (q < rs || q >= re_n) continue;
            ke[j] = s_k[q + 1];
            dst[j] = q;
            const int bs = mask_prev(s_head, q);
            u32 be = mask_next(s_head, q + 1, FN_WORDS, re_n); if (be > re_n) be = re_n;
            if (mask_any_andnot(s_diff, s_head, (u32)bs + 1, be)) {
                u32 less = 0, eqb = 0;
                for (int f = bs; f < (int)be; f += 4) {
                    u64 kf[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) kf[u] = s_k[(f + u < (int)be ? f + u : bs) + 1];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool in = f + u < (int)be;
                        less += (in && kf[u] < ke[j]) ? 1u : 0u;
                        eqb += (in && kf[u] == ke[j] && f + u < (int)q) ? 1u : 0u;
                    }
                }
                dst[j] = (u32)bs + less + eqb;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < FN_PPT; ++j) if (dst[j] != 0xFFFFFFFFu) s_k[dst[j] + 1] = ke[j];
        __syncthreads();
    }

    // ---- runs of equal keys in [rs, re) ---------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < FN_PPT; ++j) {
        const u32 q = j * FN_THREADS + tid;
        const bool r = (q >= rs && q < re) && (q == rs || s_k[q + 1] != s_k[q]);
        const u64 m = __ballot(r);
        if (lane == 0) s_run[j * 4 + wave] = m;
    }
    __syncthreads();
    u32 runlen[FN_PPT];
#pragma unroll
    for (int j = 0; j < FN_PPT; ++j) {
        const u32 q = j * FN_THREADS + tid;
        u32 c = 0;
        if (q >= rs && q < re && ((s_run[q >> 6] >> (q & 63)) & 1)) {
            u32 nx = mask_next(s_run, q + 1, FN_WORDS, re); if (nx > re) nx = re;
            c = nx - q;
            if (tail_giant && nx == re) {                                // the run reaches the end of what is staged: walk on in HBM
                const u64 me = s_k[q + 1];
                u64 g = base + nl;
                const u32 lim = a.upper < FN_MAX_WALK ? a.upper : FN_MAX_WALK;
                while (g < a.n && c <= lim && a.keys[g] == me) { ++g; ++c; }
                if (c > lim && a.upper > FN_MAX_WALK) atomicOr(a.flags, (u32)FN_FLAG_LONG_WALK);
            }
        }
        const bool keep = c >= a.lower && c <= a.upper;
        runlen[j] = keep ? c : 0;
        const u64 km = __ballot(keep);
        if (lane == 0) s_keep[j * 4 + wave] = km;
    }
    __syncthreads();
    if (tid < 64) {
        const u32 v = tid < FN_WORDS ? (u32)__popcll(s_keep[tid]) : 0;
        const u32 inc = wave_incl_scan<u32>(v);
        if (tid < FN_WORDS) s_pre[tid] = inc - v;
        if (tid == FN_WORDS - 1) s_pre[FN_WORDS] = inc;
    }
    __syncthreads();
    u32 nkept = s_pre[FN_WORDS];
    if (nkept > a.cap_t) nkept = a.cap_t;                                // cannot happen: (2048+512)/L bounds it
#pragma unroll
    for (int j = 0; j < FN_PPT; ++j) {
        const u32 c = runlen[j];
        if (!c) continue;
        const u32 q = j * FN_THREADS + tid;
        const u32 w = j * 4 + wave;
        const u32 slot = s_pre[w] + (u32)__popcll(s_keep[w] & ((1ULL << lane) - 1));
        s_opos[slot] = (u16)q; s_ocnt[slot] = (u16)c;
    }
    __syncthreads();
    if (tid == 0) a.tile_cnt[tile] = nkept;
    u64 *out = a.scratch + tile * (u64)a.cap_t * 2;
    for (u32 i = tid; i < nkept * 2; i += FN_THREADS) {
        const u32 e = i >> 1;
        out[i] = (i & 1) ? (u64)s_ocnt[e] : s_k[(u32)s_opos[e] + 1];
    }
}

// Moves the kept entries from the per-tile slots to their final place (tile_off = exclusive scan of the tile
// counts, tile_off[ntiles] = total) and builds the count histogram (print_kmer_histogram, reference
// src/hysortk.cpp:98-136).  Persistent workgroups, one histogram flush each.
__global__ __launch_bounds__(FN_THREADS) void finish_compact_kernel(const u64 *scratch, u32 cap_t, const u64 *tile_off, u64 ntiles,
                                                                     u64 *entries, u64 *histo, u32 histo_len)
{
    __shared__ u32 s_hist[FN_LDS_HIST];
    for (int i = threadIdx.x; i < FN_LDS_HIST; i += FN_THREADS) s_hist[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // one wave per tile: a tile keeps ~80 entries = 160 words
    for (u64 t = (u64)blockIdx.x * 4 + wave; t < ntiles; t += (u64)gridDim.x * 4) {
        const u64 o = tile_off[t];
        const u32 cnt = (u32)(tile_off[t + 1] - o);
        const u64 *src = scratch + t * (u64)cap_t * 2;
        for (u32 i = lane; i < cnt * 2; i += 64) {
            const u64 v = src[i];
            entries[o * 2 + i] = v;
            if (i & 1) {
                if (v < (u64)FN_LDS_HIST) atomicAdd(&s_hist[(u32)v], 1u);
                else if (v < histo_len) atomicAdd((unsigned long long *)&histo[v], 1ULL);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FN_LDS_HIST; i += FN_THREADS) {
        const u32 c = s_hist[i];
        if (c && (u32)i < histo_len) atomicAdd((unsigned long long *)&histo[i], (unsigned long long)c);
    }
}

} // namespace hsk
