// hsk_sort.h -- LSD radix sort of k-mer records (device), replacing sort_task (reference
// src/kmerops.cpp:1382-1407) and the vendored sorters it calls (dependency/Raduls raduls.h:1061,
// dependency/Paradis paradissort.hpp:211).  Same contract as the RADULS call: records of NW
// 64-bit words are ordered by their key bytes read as ONE little-endian integer (word NW-1 most
// significant; for K <= 32 plain ascending uint64), optional 8-byte payload carried along.
//
// Structure ("one sweep" per digit, written for wave64 / 160 KB LDS):
//   * hist_kernel:   ONE read of the keys builds the digit histograms of every pass
//                    (LDS-privatised per workgroup, merged with global atomics).
//   * onesweep_kernel (the dominant kernel, 2*record bytes of HBM traffic per key and pass):
//       - tiles are claimed through an atomic ticket, so a tile only ever waits on tiles whose
//         workgroups are already resident (forward progress without co-residency assumptions);
//       - keys are loaded wave-striped (coalesced 512 B per wave instruction), ranked inside the
//         wave with ballot "match" (RADIX_BITS ballots per key, stable), wave totals go through
//         wave-private LDS counters, digits are prefix-scanned across the 4 waves in LDS;
//       - the tile's digit counts are published to a look-back table as single-word
//         {flag,value} granules with agent-scope relaxed atomics (the data is the flag: no
//         separate fence; per-XCD L2s are not coherent so plain loads would be stale);
//       - while predecessors are polled the keys are permuted through LDS into digit order, so
//         the final stores are runs of consecutive addresses per digit (coalesced scatter).
//     Every spin is bounded; on timeout an error word is set and the host reports HSK_ERR_INTERNAL.
#pragma once
#include "hsk_device.h"

namespace hsk {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / WAVE;

#ifndef HSK_SORT_KPT1
#define HSK_SORT_KPT1 16
#endif
template <int NW> struct SortTile { static constexpr int KPT = (NW == 1 ? HSK_SORT_KPT1 : 16 / NW); static constexpr int TILE = SORT_THREADS * KPT; };
template <> struct SortTile<3> { static constexpr int KPT = 5; static constexpr int TILE = SORT_THREADS * 5; };


struct HistArgs {
    const u64 *keys; u64 n; int npass; PassDesc pass[MAX_PASSES];
    u64 *ghist;           // [npass][256]
};

template <int NW>
__global__ __launch_bounds__(SORT_THREADS) void hist_kernel(HistArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_h[];   // [npass][256]
    for (int i = threadIdx.x; i < a.npass * 256; i += SORT_THREADS) s_h[i] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * SORT_THREADS;
    for (u64 g = (u64)blockIdx.x * SORT_THREADS + threadIdx.x; g < a.n; g += stride) {
        u64 w[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = a.keys[g * NW + i];
        for (int p = 0; p < a.npass; ++p) {
            const PassDesc pd = a.pass[p];
            u32 d = (u32)(pick_word<NW>(w, pd.word) >> pd.shift) & ((1u << pd.bits) - 1);
            atomicAdd(&s_h[p * 256 + d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.npass * 256; i += SORT_THREADS) {
        u32 c = s_h[i];
        if (c) atomicAdd((unsigned long long *)&a.ghist[i], (unsigned long long)c);
    }
}

// look-back word: top two bits = status, rest = count
template <typename LB> struct LBT;
template <> struct LBT<u32> { static constexpr u32 AGG = 1u << 30, INCL = 2u << 30, VMASK = (1u << 30) - 1, FMASK = 3u << 30; };
template <> struct LBT<u64> { static constexpr u64 AGG = 1ULL << 62, INCL = 2ULL << 62, VMASK = (1ULL << 62) - 1, FMASK = 3ULL << 62; };

struct SortArgs {
    const u64 *keys_in; u64 *keys_out;
    const u64 *vals_in; u64 *vals_out;
    u64 n;
    u64 ntiles;           // tiles of this task and pass
    int word, shift, bits;
    int unstable;         // 1: equal digits may leave in any order (first pass of a prefix plan that ends in an aggregation)
    const u64 *gbase;     // [256] exclusive digit offsets of this pass
    void *lookback;       // [ntiles][256] LB words, zeroed
    u32 *ticket;          // zeroed
    u32 *err;             // sticky error word
    const u64 *tile_src;  // null: tile t is keys_in[t * TILE ...]; else tile t = (chunk << 32 | keys) of a chunked input (hsk_scatter.h)
    const u32 *ntiles_dev; // non-null: the tile count lives in device memory (written by chunk_tiles_kernel); `ntiles` is then an upper bound
};

constexpr u32 LOOKBACK_SPIN_LIMIT = 1u << 22;
constexpr int LB_WIN = 4;                 // predecessors inspected per look-back step (2..4 measure the same, 8 is 2 % and 16 is 5 % slower)

// Diagnostic build only (-DHSK_DIAG): per-phase shader-clock sums of the onesweep kernel, one
// stamp set per workgroup (thread 0).  Never compiled into the product library.
#ifdef HSK_DIAG
__device__ unsigned long long g_diag[32];
#define DIAG_STAMP(i) do { if (threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); diag_t[i] = t_; } } while (0)
#else
#define DIAG_STAMP(i) do { } while (0)
#endif

// One tile of one pass.  LOCAL = the look-back words of this sort are only ever touched by workgroups of
// ONE XCD (onesweep_multi_kernel): they are published with plain stores (which stay in that XCD's L2) and
// polled with L1-bypassing loads, so the flag round trip never leaves the XCD.  !LOCAL = any placement:
// agent-scope (sc1, write-through) stores and loads.
template <int NW, bool HAS_VAL, typename LB, bool LOCAL>
__device__ __forceinline__ void onesweep_tile(const SortArgs &a)
{
    constexpr int KPT = SortTile<NW>::KPT;
    constexpr int TILE = SortTile<NW>::TILE;
    typedef LBT<LB> L;
    __shared__ u64 s_keys[TILE * NW];
    __shared__ u64 s_vals[HAS_VAL ? TILE : 2];
    __shared__ u32 s_whist[SORT_WAVES * 256];
    __shared__ u32 s_dstart[256];
    __shared__ long long s_delta[256];
    __shared__ u32 s_scan[8];
    __shared__ u32 s_tile[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef HSK_DIAG
    unsigned long long diag_t[10];
#endif
    DIAG_STAMP(0);
    if (tid == 0) s_tile[0] = atomicAdd(a.ticket, 1u);
    for (int i = tid; i < SORT_WAVES * 256; i += SORT_THREADS) { s_whist[i] = 0; s_keys[i] = 0; }
    __syncthreads();
    DIAG_STAMP(1);
    const u64 tile = s_tile[0];
    const u64 ntiles_now = a.ntiles_dev ? (u64)*a.ntiles_dev : a.ntiles;
    if (tile >= ntiles_now) return;               // uniform: the whole workgroup leaves (multi kernel: task exhausted)
    u64 base = tile * TILE;
    u32 nvalid = (u32)((a.n - base) < (u64)TILE ? (a.n - base) : (u64)TILE);
    if (a.tile_src) { const u64 ts = a.tile_src[tile]; base = (ts >> 32) * TILE; nvalid = (u32)ts; }
    const u32 dmask = (1u << a.bits) - 1;

    // ---- load (wave-striped) -----------------------------------------------------------------
    u64 key[KPT][NW];
    u64 val[HAS_VAL ? KPT : 1];
    u32 dig[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const u32 idx = wave * (WAVE * KPT) + j * WAVE + lane;
        const bool ok = idx < nvalid;
#pragma unroll
        for (int w = 0; w < NW; ++w) key[j][w] = ok ? a.keys_in[(base + idx) * NW + w] : ~0ULL;
        if (HAS_VAL) val[j] = ok ? a.vals_in[base + idx] : 0;
        dig[j] = (u32)(pick_word<NW>(key[j], a.word) >> a.shift) & dmask;
    }

#ifdef HSK_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    DIAG_STAMP(2);
    // ---- rank inside the wave (stable) ----------------------------------------------------------
    // "match" without ballots: every lane ORs its lane bit into a wave-private 64-bit word per digit
    // (LDS atomic OR: commutative, so the result does not depend on the order lanes are served in),
    // reads the word back = the set of lanes holding the same digit, and ranks itself by popcount.
    // The words alias the key staging area, which is not written before the permute phase.
    u32 *wh = s_whist + wave * 256;
    u64 *wm = s_keys + wave * 256;
    const u64 lane_bit = 1ULL << lane;
    const u64 lt_mask = lane_bit - 1;
    u32 rank[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const u32 d = dig[j];
        // padding of a partial tile is not counted (a chunked input has a partial tile per digit, not only at the end)
        if ((u32)(wave * (WAVE * KPT) + j * WAVE + lane) >= nvalid) { rank[j] = 0; continue; }
        if (a.unstable) {                            // (uniform) one LDS atomic: the rank is the arrival order inside the wave
            rank[j] = __hip_atomic_fetch_add(&wh[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            continue;
        }
        __hip_atomic_fetch_or(&wm[d], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const u64 peers = __hip_atomic_load(&wm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const u32 old = __hip_atomic_load(&wh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const u32 before = (u32)__popcll(peers & lt_mask);
        const u32 cnt = (u32)__popcll(peers);
        if (before == 0) { __hip_atomic_store(&wm[d], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); __hip_atomic_store(&wh[d], old + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        rank[j] = old + before;
    }
    DIAG_STAMP(3);
    __syncthreads();
    DIAG_STAMP(4);

    // ---- digit totals, cross-wave exclusive prefix, digit start inside the tile ----------------
    u32 total;
    {
        u32 run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) { u32 c = s_whist[w * 256 + tid]; s_whist[w * 256 + tid] = run; run += c; }
        total = run;
    }
    u32 tile_total;
    const u32 dstart = block_excl_scan_256<u32>(total, s_scan, &tile_total);
    s_dstart[tid] = dstart;

    // ---- publish this tile's digit count ---------------------------------------------------------
    typedef __attribute__((address_space(1))) LB GLB;          // global (not flat) accesses for the look-back words
    GLB *lb = (GLB *)a.lookback;
    GLB *mine = lb + tile * 256 + tid;
    {
        const LB v0 = (LB)((tile == 0 ? L::INCL : L::AGG) | (LB)total);
        if (LOCAL) __hip_atomic_store(mine, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(mine, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    // ---- permute through LDS into digit order (overlaps the predecessors' progress) ------------
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const u32 d = dig[j];
        if ((u32)(wave * (WAVE * KPT) + j * WAVE + lane) >= nvalid) continue;
        const u32 lpos = s_dstart[d] + s_whist[wave * 256 + d] + rank[j];
#pragma unroll
        for (int w = 0; w < NW; ++w) s_keys[lpos * NW + w] = key[j][w];
        if (HAS_VAL) s_vals[lpos] = val[j];
    }

    DIAG_STAMP(5);
    // ---- decoupled look-back: exclusive prefix of digit `tid` over all earlier tiles ----------
    u64 excl = 0;
    if (tile > 0) {
        // A window of LB_WIN predecessors is fetched with independent loads per step (the walk is
        // latency-bound: with ~1000 tiles in flight most predecessors only carry an aggregate, and
        // a one-load-per-step walk serialises hundreds of L2 round trips).
        long long t = (long long)tile - 1;
        u32 spins = 0;
        bool done = false;
#ifdef HSK_DIAG
        u32 dg_steps = 0;
#endif
        while (!done) {
#ifdef HSK_DIAG
            ++dg_steps;
#endif
            LB v[LB_WIN];
#pragma unroll
            for (int i = 0; i < LB_WIN; ++i)
                v[i] = (t - i >= 0) ? __hip_atomic_load(lb + (u64)(t - i) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (LB)L::INCL;
            int used = 0;
#pragma unroll
            for (int i = 0; i < LB_WIN; ++i) {
                if (done || used < i) continue;              // stop at the first entry that is not ready
                const LB f = v[i] & L::FMASK;
                if (f == 0) continue;
                excl += (u64)(v[i] & L::VMASK);
                ++used;
                if (f == L::INCL) done = true;
            }
            t -= used;
            if (!done && used < LB_WIN) {                    // ran into an unpublished entry: back off and retry from it
                if (++spins > LOOKBACK_SPIN_LIMIT) { atomicOr(a.err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (LOCAL) __hip_atomic_store(mine, (LB)(L::INCL | (LB)(excl + total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(mine, (LB)(L::INCL | (LB)(excl + total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef HSK_DIAG
        if (tid == 0) { atomicAdd(&g_diag[10], (unsigned long long)dg_steps); atomicAdd(&g_diag[11], (unsigned long long)spins); atomicAdd(&g_diag[12], (unsigned long long)(tile - 1 - t)); }
#endif
    }
    s_delta[tid] = (long long)(a.gbase[tid] + excl) - (long long)dstart;
    DIAG_STAMP(6);
    __syncthreads();
    DIAG_STAMP(7);

    // ---- coalesced scatter -----------------------------------------------------------------------
    for (u32 i = tid; i < nvalid; i += SORT_THREADS) {
        u64 k[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = s_keys[i * NW + w];
        const u32 d = (u32)(pick_word<NW>(k, a.word) >> a.shift) & dmask;
        const u64 o = (u64)(s_delta[d] + (long long)i);
#pragma unroll
        for (int w = 0; w < NW; ++w) a.keys_out[o * NW + w] = k[w];
        if (HAS_VAL) a.vals_out[o] = s_vals[i];
    }
#ifdef HSK_DIAG
    DIAG_STAMP(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DIAG_STAMP(9);
    if (tid == 0) {
        for (int i = 0; i < 9; ++i) atomicAdd(&g_diag[i], diag_t[i + 1] - diag_t[i]);
        atomicAdd(&g_diag[16], 1ULL);
    }
#endif
}

// one task, any placement: one tile per workgroup
template <int NW, bool HAS_VAL, typename LB>
__global__ __launch_bounds__(SORT_THREADS) void onesweep_kernel(SortArgs a)
{
    onesweep_tile<NW, HAS_VAL, LB, false>(a);
}

// Eight independent tasks in one launch, one per XCD: a workgroup reads the id of the XCD it runs on
// (HW_REG_XCC_ID) and only ever works on that XCD's task, so every look-back word is shared inside one
// XCD (-> LOCAL flags).  The grid has (8 x the largest tile count) + 12 % workgroups; the dispatcher deals
// them round-robin over the XCDs, a workgroup whose task has no tile left exits at once, and the host
// verifies afterwards that every ticket counter passed its tile count (HSK_ERR_INTERNAL otherwise: the
// results never silently depend on the placement).
struct MultiSortArgs { SortArgs t[8]; };
constexpr unsigned XCC_ID_GETREG = 20u | (0u << 6) | (3u << 11);     // HW_REG_XCC_ID, bits [3:0]

// Census for hsk_init: how many workgroups of a plain launch land on each XCC id.  The batch kernels (one task per
// XCD) need all eight ids to show up; on a partitioned device (CPX / DPX / QPX), under a CU mask or on any other
// placement they are switched off and every task takes the placement-independent single-task kernels.
__global__ void xcc_census_kernel(u32 *cnt)
{
    if (threadIdx.x == 0) atomicAdd(&cnt[__builtin_amdgcn_s_getreg(XCC_ID_GETREG) & 15u], 1u);
}

template <int NW, bool HAS_VAL, typename LB>
__global__ __launch_bounds__(SORT_THREADS) void onesweep_multi_kernel(MultiSortArgs m)
{
    const u32 xcc = __builtin_amdgcn_s_getreg(XCC_ID_GETREG) & 7u;
    onesweep_tile<NW, HAS_VAL, LB, true>(m.t[xcc]);       // leaves at once when the XCD's task has no tile left
}



// ------------------------------------------------------------------------------------------------------
// Hybrid finish for one-word keys: after LSD passes over the TOP bits only (hi_shift..63) the array is
// ordered by that prefix; equal keys share a prefix "bin", bins are short (a few tens of records at the
// task sizes used), so the remaining low bits are ordered INSIDE each bin in LDS instead of by 5 more
// global passes (16 B of HBM traffic per key each).  Every workgroup handles the records at ITS tile
// positions; a record finds its bin [bs, be) with bit scans over a head mask (both directions, the tile
// is staged with a halo on both sides) and its place by counting smaller / equal-and-earlier records of
// the bin -- or keeps its place when the bin holds one distinct key (the common case: duplicates of one
// k-mer), which a second mask decides without touching the records.  Bins longer than the halo ("giant")
// are passed through unchanged; if any of them holds two different keys a flag is raised and the host
// finishes that task with the ordinary full-width passes.  Out of place: every output slot is written once.
// ------------------------------------------------------------------------------------------------------
constexpr int BS_THREADS = 256;
constexpr int BS_PPT = 8;
constexpr int BS_TILE = BS_THREADS * BS_PPT;   // 2048 records per workgroup
constexpr int BS_HALO = 512;                   // bins up to this length are ordered locally
constexpr int BS_NL = BS_TILE + 2 * BS_HALO;   // staged records
constexpr int BS_WORDS = BS_NL / 64;

struct BinSortArgs { const u64 *in; u64 *out; const u64 *vin; u64 *vout; u64 n; int hi_shift; u32 *mixed_giant; };   // vin/vout: optional payload

__global__ __launch_bounds__(BS_THREADS) void binsort_kernel(BinSortArgs a)
{
    __shared__ u64 s_k[BS_NL + 1];             // [0] = record before the staged range
    __shared__ u64 s_head[BS_WORDS + 1];       // bit q: record q starts a bin
    __shared__ u64 s_diff[BS_WORDS + 1];       // bit q: record q differs (all 64 bits) from record q-1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = (u64)blockIdx.x * BS_TILE;
    const u64 lo = base >= (u64)BS_HALO ? base - BS_HALO : 0;
    u64 hi = base + BS_TILE + BS_HALO; if (hi > a.n) hi = a.n;
    const u32 nl = (u32)(hi - lo);
    const u32 t0 = (u32)(base - lo);
    const u32 t1 = (u32)(((base + BS_TILE) < a.n ? (base + BS_TILE) : a.n) - lo);
    for (u32 i = tid; i < nl + 1; i += BS_THREADS) {
        const long long g = (long long)lo + (long long)i - 1;
        s_k[i] = g >= 0 ? a.in[g] : 0;
    }
    __syncthreads();
    for (u32 q0 = wave * 64; q0 < (u32)BS_NL; q0 += BS_THREADS) {           // masks over the whole staged range
        const u32 q = q0 + lane;
        bool h = false, d = false;
        if (q < nl) {
            const u64 k = s_k[q + 1], kp = s_k[q];
            const bool first = (lo + q == 0);
            h = first || ((k >> a.hi_shift) != (kp >> a.hi_shift));
            d = first || (k != kp);
        }
        const u64 mh = __ballot(h), md = __ballot(d);
        if (lane == 0) { s_head[q0 >> 6] = mh; s_diff[q0 >> 6] = md; }
    }
    __syncthreads();
    const bool at_end = (hi == a.n);
    // fast path: no record of the staged range differs from its predecessor inside a bin -> every bin that
    // touches this tile holds one key -> the tile is already in order
    {
        __shared__ u32 s_any;
        if (tid == 0) s_any = 0;
        __syncthreads();
        if (tid < BS_WORDS && (s_diff[tid] & ~s_head[tid]) != 0) s_any = 1;
        __syncthreads();
        if (s_any == 0) {
#pragma unroll
            for (int j = 0; j < BS_PPT; ++j) {
                const u32 q = t0 + j * BS_THREADS + tid;
                if (q < t1) { a.out[lo + q] = s_k[q + 1]; if (a.vin) a.vout[lo + q] = a.vin[lo + q]; }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < BS_PPT; ++j) {
        const u32 q = t0 + j * BS_THREADS + tid;
        if (q >= t1) continue;
        const u64 ke = s_k[q + 1];
        // bin start: last head at or before q
        int bs = -1;
        {
            int w = (int)(q >> 6);
            u64 m = s_head[w] & (((q & 63) == 63) ? ~0ULL : ((2ULL << (q & 63)) - 1));
            while (m == 0 && w > 0) m = s_head[--w];
            if (m) bs = w * 64 + 63 - __builtin_clzll(m);
        }
        // bin end: next head after q (or the end of the array)
        int be = -1;
        {
            u32 w = q >> 6;
            u64 m = s_head[w] & (((q & 63) == 63) ? 0ULL : (~0ULL << ((q & 63) + 1)));
            while (m == 0 && ++w < (u32)BS_WORDS) m = s_head[w];
            if (m && (w * 64 + (u32)__builtin_ctzll(m)) < nl) be = (int)(w * 64 + (u32)__builtin_ctzll(m));
            else if (at_end) be = (int)nl;
        }
        u64 dst = lo + q;
        if (bs < 0 || be < 0 || (be - bs) > BS_HALO) {
            // giant bin: passed through; two different keys inside it -> the host must finish this task the long way
            const bool is_head = (s_head[q >> 6] >> (q & 63)) & 1;
            const bool differs = (s_diff[q >> 6] >> (q & 63)) & 1;
            if (differs && !is_head) atomicOr(a.mixed_giant, 1u);
        } else {
            // one distinct key in the bin? (no "differs" bit strictly inside (bs, be))
            bool uniform = true;
            {
                const u32 a0 = (u32)bs + 1, a1 = (u32)be;              // bits [a0, a1)
                if (a0 < a1) {
                    u32 w0 = a0 >> 6, w1 = (a1 - 1) >> 6;
                    for (u32 w = w0; w <= w1 && uniform; ++w) {
                        u64 m = s_diff[w];
                        if (w == w0) m &= ~0ULL << (a0 & 63);
                        if (w == w1 && ((a1 & 63) != 0)) m &= (1ULL << (a1 & 63)) - 1;
                        if (m) uniform = false;
                    }
                }
            }
            if (!uniform) {
                u32 less = 0, eqb = 0;
                for (int f = bs; f < be; f += 4) {                     // 4 independent LDS reads per step (latency bound otherwise)
                    u64 kf[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) kf[u] = s_k[(f + u < be ? f + u : bs) + 1];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool in = f + u < be;
                        less += (in && kf[u] < ke) ? 1u : 0u;
                        eqb += (in && kf[u] == ke && f + u < (int)q) ? 1u : 0u;
                    }
                }
                dst = lo + (u64)bs + less + eqb;
            }
        }
        a.out[dst] = ke;
        if (a.vin) a.vout[dst] = a.vin[lo + q];          // the payload follows its record (equal keys keep their order: eqb)
    }
}

} // namespace hsk
