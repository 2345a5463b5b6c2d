// hsk_estimate.h -- the k-mer spectrum of a SKETCH of the input, taken before the call commits to a plan (estimate_plan, hsk_api.hip).
//
// hysortk::kmer_count() is called once per process (reference src/hysortk.cpp:36-96): a plan that pays only for some inputs (the
// combining extraction of hsk_combine.h, the LDS aggregation's first table, aggregating at all) must be chosen from the input itself,
// inside the call.  What all of these depend on is ONE number: distinct canonical k-mers per k-mer instance.  The sketch:
//   * the reads that lie inside the first 1/64 of the packed buffer (4 - 64 MB: `sample`),
//   * of their canonical k-mers those whose 64-bit mix falls into a 1/32 slice of the hash space (ALL copies of a chosen k-mer inside
//     the sample are seen, so the chosen k-mers' multiplicities are exact),
//   * counted in a global open-addressing table (a few million inserts: ~0.3 ms), whose occupancy histogram gives n1, n2, n3 (k-mers
//     seen once, twice, three times), the distinct chosen k-mers and their instances.
// No minimizers, no supermers, no sort: one kernel rolls every k-mer of the sample once.  K < 64 (two-word k-mers enter the table as a
// 64-bit fingerprint of the smaller strand); the canonical form
// here is right-aligned (any consistent choice of strand does), so nothing of this is comparable with the lists the path produces.
#pragma once
#include "hsk_device.h"

namespace hsk {

constexpr int EST_SPAN = 64;                       // base positions per lane (+ K - 1 of warm-up: short spans, many lanes -- the kernel is bound by the latency of its byte loads)
constexpr int EST_THREADS = 256;
constexpr u32 EST_SELECT_BITS = 5;                 // 1 / 32 of the k-mer space
constexpr int EST_MAX_PROBES = 256;

struct EstimateArgs {
    const u8 *packed; const u64 *roff; const u32 *rlen; u64 nreads;      // the sample: reads [0, nreads), their bytes [0, roff[nreads])
    u64 positions;                                                       // 4 x sample bytes
    int k;
    unsigned long long *keys; u32 *cnts; u64 cap_mask;                   // table: key + 1 (0 = empty), count
    unsigned long long *out;                                             // [0] inserts that found no slot, [1..3] n1 n2 n3, [4] distinct, [5] instances,
                                                                         // [6] instances of the all-A k-mer (A or T runs), [7] of the all-C k-mer (C or G runs): EXACT counts inside the sample
};

__device__ __forceinline__ u64 est_find_read(const u64 *roff, u64 nreads, u64 byte)
{
    u64 lo = 0, hi = nreads - 1;                                         // last read whose offset is <= byte
    while (lo < hi) { const u64 mid = (lo + hi + 1) >> 1; if (roff[mid] <= byte) lo = mid; else hi = mid - 1; }
    return lo;
}

__device__ __forceinline__ u32 est_insert(const EstimateArgs &a, u64 key, u32 copies = 1u)      // key: the k-mer itself (K <= 32) or a 64-bit fingerprint of it; 1: no slot found
{
    if ((key * 0x9e3779b97f4a7c15ULL) >> (64 - EST_SELECT_BITS)) return 0;      // not in the slice (one multiply for 31 of 32 k-mers; the full mix only for the chosen)
    const u64 h = fmix64(key + 0x9e3779b97f4a7c15ULL);
    u64 slot = (h >> 8) & a.cap_mask;
    const unsigned long long want = key + 1ULL;
    for (int probes = 0; probes < EST_MAX_PROBES; ++probes) {
        unsigned long long prev = a.keys[slot];
        if (prev == 0ULL) prev = atomicCAS(&a.keys[slot], 0ULL, want);
        if (prev == 0ULL || prev == want) { atomicAdd(&a.cnts[slot], copies); return 0; }
        slot = (slot + 1) & a.cap_mask;
    }
    return 1;
}

// The two k-mers a read set can hold millions of times -- poly-A / poly-T tails and the poly-G reads of two-colour sequencers (canonical: all-A and
// all-C) -- are counted EXACTLY inside the sample, whatever the hash slice: a k-mer with more than U copies in the sample alone is certain to be
// dropped from the result, and the scan may leave its instances out (scan_kernel<.., DROP>).  lowmask: the bits of the k-mer's low word (K <= 32:
// of the k-mer), hi: its high word (two-word k-mers; its mask is the caller's hmask passed as lowmask of the high word -- see the callers).
__device__ __forceinline__ void est_homopolymer(const EstimateArgs &a, u64 hi, u64 lo, u64 mask_of_top, u32 copies)
{
    const bool wide = a.k > 32;
    const u64 c_lo = wide ? 0x5555555555555555ULL : (0x5555555555555555ULL & mask_of_top), c_hi = wide ? (0x5555555555555555ULL & mask_of_top) : 0ULL;
    if (hi == 0 && lo == 0) atomicAdd(&a.out[6], (unsigned long long)copies);
    else if (hi == c_hi && lo == c_lo) atomicAdd(&a.out[7], (unsigned long long)copies);
}

// two-word k-mers (32 < K < 64): both strands rolled as {hi, lo} of a right-aligned 2K-bit number, the smaller one folded to 64 bits
__device__ __forceinline__ void estimate_insert_wide(const EstimateArgs &a, u64 p0, u64 p1)
{
    const int k = a.k, hb = 2 * k - 64;                                  // bits of the high word
    const u64 hmask = (1ULL << hb) - 1ULL;
    u64 r = est_find_read(a.roff, a.nreads, p0 >> 2);
    u64 rstart = a.roff[r] * 4, rend = rstart + a.rlen[r];
    u64 nxt = r + 1 < a.nreads ? a.roff[r + 1] * 4 : ~0ULL;
    u64 fh = 0, fl = 0, rh = 0, rl = 0; u32 have = 0, lost = 0;
    u64 run_h = 0, run_l = 0; u32 run = 0;
    u64 p = p0 >= (u64)(k - 1) ? p0 - (u64)(k - 1) : 0;
    if (p < rstart) p = rstart < p0 ? rstart : p0;
    for (; p < p1; ++p) {
        while (p >= nxt) { ++r; rstart = nxt; rend = rstart + a.rlen[r]; nxt = r + 1 < a.nreads ? a.roff[r + 1] * 4 : ~0ULL; have = 0; }
        if (p < rstart || p >= rend) { have = 0; continue; }
        const u64 b = (a.packed[p >> 2] >> (6 - 2 * (u32)(p & 3))) & 3u;
        fh = ((fh << 2) | (fl >> 62)) & hmask; fl = (fl << 2) | b;
        rl = (rl >> 2) | (rh << 62); rh = (rh >> 2) | ((3ULL - b) << (hb - 2));
        if (++have < (u32)k || p < p0) continue;
        const bool rc_less = rh < fh || (rh == fh && rl < fl);
        const u64 kh = rc_less ? rh : fh, kl = rc_less ? rl : fl;
        if (run && kh == run_h && kl == run_l) { ++run; continue; }      // (runs of one k-mer as one insert, as below)
        if (run) { lost += est_insert(a, (run_l ^ fmix64(run_h + 0x632be59bd9b4e019ULL)) & ~(1ULL << 63), run); est_homopolymer(a, run_h, run_l, hmask, run); }      // (fingerprint; bit 63 cleared so that key + 1 never wraps to the empty marker)
        run_h = kh; run_l = kl; run = 1;
    }
    if (run) { lost += est_insert(a, (run_l ^ fmix64(run_h + 0x632be59bd9b4e019ULL)) & ~(1ULL << 63), run); est_homopolymer(a, run_h, run_l, hmask, run); }
    if (lost) atomicAdd(&a.out[0], (unsigned long long)lost);
}

__global__ __launch_bounds__(EST_THREADS) void estimate_insert_kernel(EstimateArgs a)
{
    const u64 t = (u64)blockIdx.x * EST_THREADS + threadIdx.x;
    const u64 p0 = t * EST_SPAN;
    if (p0 >= a.positions || a.nreads == 0) return;
    const u64 p1 = p0 + EST_SPAN < a.positions ? p0 + EST_SPAN : a.positions;
    const int k = a.k;
    if (k > 32) { estimate_insert_wide(a, p0, p1); return; }
    const u64 kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    u64 r = est_find_read(a.roff, a.nreads, p0 >> 2);
    u64 rstart = a.roff[r] * 4, rend = rstart + a.rlen[r];
    u64 nxt = r + 1 < a.nreads ? a.roff[r + 1] * 4 : ~0ULL;
    u64 fw = 0, rc = 0; u32 have = 0;                                    // rolled strands (right-aligned), bases rolled since the read began / the warm-up started
    u64 run_key = 0; u32 run = 0;                                        // the k-mer seen last and how many times in a row
    // the k-mers that END inside [p0, p1) are this lane's: warm up on the k - 1 bases before p0 (same read only)
    u64 p = p0 >= (u64)(k - 1) ? p0 - (u64)(k - 1) : 0;
    if (p < rstart) p = rstart < p0 ? rstart : p0;
    u32 lost = 0;
    for (; p < p1; ++p) {
        while (p >= nxt) { ++r; rstart = nxt; rend = rstart + a.rlen[r]; nxt = r + 1 < a.nreads ? a.roff[r + 1] * 4 : ~0ULL; have = 0; }
        if (p < rstart || p >= rend) { have = 0; continue; }             // padding behind a read's last base / a gap between reads
        const u32 b = (a.packed[p >> 2] >> (6 - 2 * (u32)(p & 3))) & 3u;
        fw = ((fw << 2) | b) & kmask;
        rc = (rc >> 2) | ((u64)(3u - b) << (2 * (k - 1)));
        if (++have < (u32)k || p < p0) continue;
        // (runs of one k-mer -- homopolymers -- enter as one insert with their length: 5 % of all-A reads made the table's one counter for
        //  that k-mer the whole sketch: 36 ms instead of 1.2)
        const u64 key = rc < fw ? rc : fw;
        if (run && key == run_key) { ++run; continue; }
        if (run) { lost += est_insert(a, run_key, run); est_homopolymer(a, 0, run_key, kmask, run); }
        run_key = key; run = 1;
    }
    if (run) { lost += est_insert(a, run_key, run); est_homopolymer(a, 0, run_key, kmask, run); }
    if (lost) atomicAdd(&a.out[0], (unsigned long long)lost);
}

__global__ __launch_bounds__(EST_THREADS) void estimate_hist_kernel(EstimateArgs a)
{
    __shared__ unsigned long long s_acc[5];
    if (threadIdx.x < 5) s_acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long n1 = 0, n2 = 0, n3 = 0, d = 0, n = 0;
    const u64 stride = (u64)gridDim.x * EST_THREADS;
    for (u64 s = (u64)blockIdx.x * EST_THREADS + threadIdx.x; s <= a.cap_mask; s += stride) {
        const u32 c = a.cnts[s];
        if (!c) continue;
        ++d; n += c; n1 += c == 1; n2 += c == 2; n3 += c == 3;
    }
    if (d) { atomicAdd(&s_acc[0], n1); atomicAdd(&s_acc[1], n2); atomicAdd(&s_acc[2], n3); atomicAdd(&s_acc[3], d); atomicAdd(&s_acc[4], n); }
    __syncthreads();
    if (threadIdx.x < 5 && s_acc[threadIdx.x]) atomicAdd(&a.out[1 + threadIdx.x], s_acc[threadIdx.x]);
}

} // namespace hsk
