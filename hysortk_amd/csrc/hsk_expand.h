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
// launches: tile sums (bytes, k-mers per EXP_TILE supermers) -> per-segment exclusive scan -> expand.
#pragma once
#include <vector>
#include "hsk_device.h"

namespace hsk {

constexpr int EXP_THREADS = 256;
constexpr int EXP_SPT = 2;
constexpr int EXP_TILE = EXP_THREADS * EXP_SPT;   // supermers per tile

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

// Tile sums and their scan for up to EXP_PREP_BATCH tasks per launch (blockIdx.y = task)
constexpr int EXP_PREP_BATCH = 8;
struct ExpandPrepArgs {
    const ExpSeg *segs[EXP_PREP_BATCH]; int nseg[EXP_PREP_BATCH]; const u8 *sm_len[EXP_PREP_BATCH];
    u64 ntiles[EXP_PREP_BATCH]; u64 *tile_sum[EXP_PREP_BATCH]; u64 *tile_off[EXP_PREP_BATCH]; int k;
};

// tile_sum[tile] = {bytes, kmers}; a wave per tile, a lane eight consecutive lengths (one 8-byte load: a lane per length left the kernel waiting
// for 64-byte requests, 0.26 ms per 100 M supermers)
constexpr int EXP_SUM_TILES = EXP_THREADS / WAVE;      // tiles per workgroup
static_assert(EXP_TILE == 8 * WAVE, "eight supermers per lane");
__global__ __launch_bounds__(EXP_THREADS) void expand_tilesum_kernel(ExpandPrepArgs a)
{
    const int ti = blockIdx.y;
    const u64 tile = (u64)blockIdx.x * EXP_SUM_TILES + (threadIdx.x >> 6);
    if (tile >= a.ntiles[ti]) return;
    const ExpSeg *segs = a.segs[ti]; const u8 *sm_len = a.sm_len[ti];
    const ExpSeg seg = segs[seg_of_tile(segs, a.nseg[ti], tile)];
    const u64 first = (tile - seg.tile_start) * EXP_TILE + 8u * (u32)lane_id();
    u32 nb = 0, nk = 0;
    if (first < seg.n_sup) {
        const u32 n_ok = seg.n_sup - first < 8 ? (u32)(seg.n_sup - first) : 8u;
        const u64 l8 = *reinterpret_cast<const u64 *>(sm_len + seg.sup_off + first);      // (unaligned; behind the segment's last length: padding or the next segment)
#pragma unroll
        for (int j = 0; j < 8; ++j) if ((u32)j < n_ok) { const u32 len = (u32)(l8 >> (8 * j)) & 255u; nb += (len + 3) >> 2; nk += len - (u32)a.k + 1u; }
    }
    for (int o = 32; o > 0; o >>= 1) { nb += __shfl_down(nb, o, WAVE); nk += __shfl_down(nk, o, WAVE); }
    if (lane_id() == 0) { a.tile_sum[ti][2 * tile] = nb; a.tile_sum[ti][2 * tile + 1] = nk; }
}

// one block per (segment, task): tile_off[tile] = {absolute byte offset, k-mer offset relative to task};
// 8 consecutive tiles per thread and step
__global__ __launch_bounds__(EXP_THREADS) void expand_scan_kernel(ExpandPrepArgs a)
{
    constexpr int IPT = 8;
    __shared__ u64 s_scr[8];
    __shared__ u64 s_carry[2];
    const int ti = blockIdx.y, sg = blockIdx.x;
    if (sg >= a.nseg[ti]) return;
    const ExpSeg seg = a.segs[ti][sg];
    const u64 *tile_sum = a.tile_sum[ti]; u64 *tile_off = a.tile_off[ti];
    const u64 t0 = seg.tile_start;
    const u64 t1 = (sg + 1 < a.nseg[ti]) ? a.segs[ti][sg + 1].tile_start : a.ntiles[ti];
    if (threadIdx.x == 0) { s_carry[0] = seg.byte_off; s_carry[1] = seg.kmer_off; }
    __syncthreads();
    for (u64 base = t0; base < t1; base += (u64)EXP_THREADS * IPT) {
        const u64 tf = base + (u64)threadIdx.x * IPT;
        u64 b[IPT], kk[IPT], sb = 0, sk = 0;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const u64 t = tf + i;
            b[i] = 0; kk[i] = 0;
            if (t < t1) { b[i] = tile_sum[2 * t]; kk[i] = tile_sum[2 * t + 1]; }
            sb += b[i]; sk += kk[i];
        }
        u64 tb, tk;
        u64 eb = block_excl_scan_256<u64>(sb, s_scr, &tb) + s_carry[0];
        u64 ek = block_excl_scan_256<u64>(sk, s_scr, &tk) + s_carry[1];
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const u64 t = tf + i;
            if (t < t1) { tile_off[2 * t] = eb; tile_off[2 * t + 1] = ek; }
            eb += b[i]; ek += kk[i];
        }
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
// 8 consecutive k-mers.  A lane takes one item (item -> supermer through a map the tile prologue writes to LDS):
// it pulls a window of the base stream once (all loads of a lane are issued together) and then ROLLS: the forward
// k-mer is the window shifted by one base per step, its twin is shifted the other way with the complement of the
// entering base on top (the reference recomputes Kmer::GetTwin per k-mer, include/kmer.hpp:266-296).
//
// Output order inside a task is free (the sort follows), so nothing is staged: the 64 items of a wave own one
// contiguous range of the task's key array (items are consecutive), and the wave fills it ROUND-MAJOR: in round r
// the lanes that still have an r-th k-mer store to consecutive slots (rank = popcount of the ballot below the
// lane).  Every store instruction of a wave writes one contiguous run of up to 512 bytes; no LDS, no barrier.
//
// One launch expands up to EXP_BATCH tasks (the batch the sort takes next).  Workgroups are persistent and the
// blockIdx -> (task, tile) map is XCD-aware: the dispatcher deals workgroups round-robin over the 8 XCDs, XCD x
// takes the tile rows x, x+8, ... of ALL tasks of the batch.  Tile row i of every task covers about the same
// stretch of the packed reads (a task's supermers are in read order), so a 128-byte line of reads that holds
// bases of several tasks is fetched from HBM once per batch and hit in that XCD's L2 by the other tasks
// (one launch per task fetched nearly the whole read set once per TASK: 87 GB for 64 GB of keys written).
//
// While the keys are in registers the digit histograms of the radix passes that follow are counted
// (LDS-privatised, merged with global atomics once per workgroup): the sort needs no histogram pass of its own.
constexpr int EXP_RUN = 8;                            // k-mers per work item
constexpr int EXP_MAX_ITEMS = EXP_TILE * 16;          // a supermer has at most 128 k-mers (SUPERMER_CUT)
constexpr int EXP_BATCH = 8;

struct ExpandTask {
    const ExpSeg *segs; int nseg; u32 pad_;
    const u8 *sm_len; const u64 *src8; u64 src_bit0, src_words;
    const u64 *sm_gpos; const u32 *sm_pos; const int32_t *sm_rid;
    const u32 *sm_boff;            // byte-store mode: supermer s starts at byte seg.byte_off + sm_boff[s] of src8 (src_bit0 = 0)
    const u64 *tile_off; u64 ntiles;
    u64 *keys_out, *vals_out;
    u64 *ghist;                    // [npass][256] digit histograms of this task (null: none)
    u64 *kcursor;                  // tile_off == null (bases read in place, one segment): next free record of the task's key array;
                                   // a tile reserves its range with one atomic (the order of the k-mers inside a task is free)
};
struct ExpandArgs { ExpandTask t[EXP_BATCH]; int ntask, k; u32 row_workers; int npass; u64 nrows; PassDesc pass[MAX_PASSES]; };

template <int NW, bool EXT>
__global__ __launch_bounds__(EXP_THREADS) void expand_kernel(ExpandArgs a)
{
    __shared__ u32 s_boff[EXP_TILE + 1];
    __shared__ u32 s_koff[EXP_TILE + 1];
    __shared__ u32 s_ioff[EXP_TILE + 1];
    __shared__ u16 s_isup[EXP_MAX_ITEMS];
    __shared__ u64 s_gpos[EXP_TILE];                                  // first base of every supermer of the tile (reference mode)
    __shared__ u32 s_scr[8];
    __shared__ u64 s_kb;
    extern __shared__ __attribute__((aligned(16))) u32 s_hist[];       // [npass][256]
    const int tid = threadIdx.x;
    const u32 xcd = blockIdx.x & 7u, wk = blockIdx.x >> 3;
    const u32 ti = wk % (u32)a.ntask, row0 = wk / (u32)a.ntask;
    const ExpandTask &t = a.t[ti];
    const int k = a.k;
    const bool do_hist = t.ghist != nullptr && a.npass > 0;
    if (do_hist) for (int i = tid; i < a.npass * 256; i += EXP_THREADS) s_hist[i] = 0;
    const int low = 64 * NW - 2 * k;                      // unused low bits of the last word; 0 < low < 64 (k % 32 != 0)
    const u64 lastmask = ~0ULL << low;
    __syncthreads();

    // rows are PROPORTIONAL positions in the tasks' tile lists: row r of a task with n tiles is the tiles
    // [r * n / nrows, (r + 1) * n / nrows), so that one row of every task reads the same stretch of the packed reads
    // even though the tasks differ in size by a few percent (nrows = the largest tile count of the batch)
    for (u64 row = xcd + 8ULL * row0; row < a.nrows; row += 8ULL * a.row_workers)
    for (u64 tile = row * t.ntiles / a.nrows, tile_hi = (row + 1) * t.ntiles / a.nrows; tile < tile_hi; ++tile) {
        const int sg = seg_of_tile(t.segs, t.nseg, tile);
        const ExpSeg seg = t.segs[sg];
        const u64 first = (tile - seg.tile_start) * EXP_TILE;
        const u32 ns = (u32)((seg.n_sup - first) < (u64)EXP_TILE ? (seg.n_sup - first) : (u64)EXP_TILE);

        // ---- prologue: blocked arrangement, thread t owns supermers [t*SPT, t*SPT+SPT) of the tile -------------
        u32 nb[EXP_SPT], nk[EXP_SPT], sb = 0, sk = 0, si = 0;
#pragma unroll
        for (int i = 0; i < EXP_SPT; ++i) {
            const u32 s = tid * EXP_SPT + i;
            const u32 len = (s < ns) ? t.sm_len[seg.sup_off + first + s] : 0;
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
            const u32 s = tid * EXP_SPT + i;
            const u32 ni = (nk[i] + EXP_RUN - 1) / EXP_RUN;
            s_boff[s] = eb; s_koff[s] = ek; s_ioff[s] = ei;
            if (s < ns) {
                if (t.sm_boff) s_gpos[s] = 4 * (seg.byte_off + (u64)t.sm_boff[seg.sup_off + first + s]);
                else if (t.sm_gpos) s_gpos[s] = t.sm_gpos[seg.sup_off + first + s];
            }
            for (u32 j = 0; j < ni; ++j) s_isup[ei + j] = (u16)s;        // item -> supermer
            eb += nb[i]; ek += nk[i]; ei += ni;
        }
        if (tid == EXP_THREADS - 1) { s_boff[EXP_TILE] = eb; s_koff[EXP_TILE] = ek; s_ioff[EXP_TILE] = ei; }
        __syncthreads();

        u64 byte_abs = 0, kbase;
        if (t.tile_off) { byte_abs = t.tile_off[2 * tile]; kbase = t.tile_off[2 * tile + 1]; }
        else {
            if (tid == 0) s_kb = atomicAdd((unsigned long long *)t.kcursor, (unsigned long long)totk);
            __syncthreads();
            kbase = s_kb;
        }

        // The window of step s+1 is requested before step s is computed (two dependent memory latencies per step
        // -- item -> position -> bases -- would otherwise sit in front of every 8 rounds of arithmetic).
        u32 n_cnt = 0, n_out0 = 0, n_sh = 0; u64 n_vbase = 0; u64 n_raw[NW + 2];
        auto fetch = [&](u32 item) {
            n_cnt = 0; n_out0 = 0; n_sh = 0; n_vbase = 0;
#pragma unroll
            for (int x = 0; x < NW + 2; ++x) n_raw[x] = 0;
            if (item < toti) {
                const u32 sidx = s_isup[item];
                const u32 i0 = (item - s_ioff[sidx]) * EXP_RUN;
                const u32 k0 = s_koff[sidx];
                const u32 nks = s_koff[sidx + 1] - k0;
                n_cnt = nks - i0 < (u32)EXP_RUN ? nks - i0 : (u32)EXP_RUN;
                n_out0 = k0 + i0;
                const u64 bit = (t.sm_gpos || t.sm_boff) ? (t.src_bit0 + 2 * (s_gpos[sidx] + (u64)i0)) : (8 * (byte_abs + s_boff[sidx]) + 2 * (u64)i0);
                const u64 wi = bit >> 6; n_sh = (u32)(bit & 63);
#pragma unroll
                for (int x = 0; x < NW + 2; ++x) n_raw[x] = (wi + x < t.src_words) ? t.src8[wi + x] : 0;
                if (EXT) { const u64 sabs = seg.sup_off + first + sidx; n_vbase = (u64)(t.sm_pos[sabs] + i0) | ((u64)(u32)t.sm_rid[sabs] << 32); }
            }
        };
        fetch(tid);
        for (u32 it0 = 0; it0 < toti; it0 += EXP_THREADS) {
            const u32 cnt = n_cnt, out0 = n_out0; const u64 vbase = n_vbase;
            u64 win[NW + 1];
            {
                u64 aw[NW + 2];
#pragma unroll
                for (int x = 0; x < NW + 2; ++x) aw[x] = __builtin_bswap64(n_raw[x]);
#pragma unroll
                for (int x = 0; x < NW + 1; ++x) win[x] = n_sh ? ((aw[x] << n_sh) | (aw[x + 1] >> (64 - n_sh))) : aw[x];
            }
            if (it0 + EXP_THREADS < toti) fetch(it0 + EXP_THREADS + tid);
            Mer<NW> fw, rc;
#pragma unroll
            for (int x = 0; x < NW; ++x) fw.w[x] = win[x];
            fw.w[NW - 1] &= lastmask;
            rc = twin<NW>(fw, k);
            // the wave's items are consecutive: their k-mers fill [wbase, wbase + sum of cnt) of the tile's range
            const u32 wbase = (u32)__shfl((int)out0, 0, WAVE);
            u32 run = 0;
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
                const bool act = (u32)r < cnt;
                const u64 m = __ballot(act);
                if (m == 0) break;                                    // uniform: no lane of the wave has an r-th k-mer
                if (act) {
                    const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
                    const u64 o = kbase + wbase + run + below;
                    const bool use_rc = mer_less<NW>(rc, fw);
                    u64 key[NW];
#pragma unroll
                    for (int x = 0; x < NW; ++x) key[x] = use_rc ? rc.w[x] : fw.w[x];
                    if (NW == 2) *reinterpret_cast<ulonglong2 *>(t.keys_out + o * 2) = make_ulonglong2(key[0], key[NW - 1]);
                    else {
#pragma unroll
                        for (int x = 0; x < NW; ++x) t.keys_out[o * NW + x] = key[x];
                    }
                    if (EXT) t.vals_out[o] = vbase + (u64)r;          // pos + r (the low word never carries: pos < read length)
                    if (do_hist) {
                        for (int p = 0; p < a.npass; ++p) {
                            const PassDesc pd = a.pass[p];
                            const u32 d = (u32)(pick_word<NW>(key, pd.word) >> pd.shift) & ((1u << pd.bits) - 1);
                            atomicAdd(&s_hist[p * 256 + d], 1u);
                        }
                    }
                }
                run += (u32)__popcll(m);
            }
        }
        __syncthreads();                                              // the next tile's prologue rewrites the LDS tables
    }
    if (do_hist) {
        __syncthreads();
        for (int i = tid; i < a.npass * 256; i += EXP_THREADS) {
            const u32 cv = s_hist[i];
            if (cv) atomicAdd((unsigned long long *)&t.ghist[i], (unsigned long long)cv);
        }
    }
}

// Multi-GPU only: materialise the re-aligned byte stream of reference-mode supermers for the exchange (what
// SupermerEncoder::copy_bits writes in the reference, src/kmerops.cpp:1096-1107; `(len + 3) / 4` bytes per supermer,
// tail bits zero).  One lane per aligned 8-byte word of the OUTPUT: the supermer under the word's first byte is found
// by binary search over the tile's byte prefix sums, then the word is assembled from one or two (for tiny K: a few)
// supermers, each piece one unaligned 64-bit pull from the packed reads.  The bytes of a tile before its first and
// behind its last aligned word are written one by one (the neighbouring tiles write the rest of those words).
__device__ __forceinline__ u64 pack_pull(const u32 *s_boff, const u64 *s_gpos, const u8 *s_len, u32 ns, u32 o, u32 n,
                                         const u64 *src8, u64 src_bit0, u64 src_words)
{
    u32 lo = 0, hi = ns - 1;                                      // supermer under tile-relative byte o
    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_boff[mid] <= o) lo = mid; else hi = mid - 1; }
    u64 r = 0;
    u32 i = 0;
    while (i < n && lo < ns) {
        const u32 jb = o + i - s_boff[lo];                        // first byte wanted inside supermer `lo`
        const u32 nbs = s_boff[lo + 1] - s_boff[lo];
        u32 take = nbs - jb; if (take > n - i) take = n - i;
        const u64 bits = bits64_bytes_clamped(src8, src_bit0 + 2 * s_gpos[lo] + 8 * (u64)jb, src_words);
        u64 le = __builtin_bswap64(bits);                         // byte j of `le` = j-th byte of the stream
        const u32 len = s_len[lo];
        if ((len & 3) && jb + take == nbs) le &= ~((u64)(0xFFu >> (2 * (len & 3))) << (8 * (take - 1)));   // tail bits of the supermer's last byte
        if (take < 8) le &= (1ULL << (8 * take)) - 1;
        r |= le << (8 * i);
        i += take; ++lo;
    }
    return r;
}

__global__ __launch_bounds__(EXP_THREADS) void pack_kernel(const ExpSeg *segs, int nseg, const u8 *sm_len, const u64 *src8, u64 src_bit0, u64 src_words,
                                                            const u64 *sm_gpos, const u64 *tile_off, u8 *bytes_out)
{
    __shared__ u32 s_boff[EXP_TILE + 1];
    __shared__ u64 s_gpos[EXP_TILE];
    __shared__ u8 s_len[EXP_TILE];
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
        const u32 sidx = tid * EXP_SPT + i;
        const u32 len = (sidx < ns) ? sm_len[seg.sup_off + first + sidx] : 0;
        nb[i] = (sidx < ns) ? ((len + 3) >> 2) : 0;
        sb += nb[i];
        s_len[sidx] = (u8)len;
        if (sidx < ns) s_gpos[sidx] = sm_gpos[seg.sup_off + first + sidx];
    }
    u32 totb;
    u32 eb = block_excl_scan_256<u32>(sb, s_scr, &totb);
#pragma unroll
    for (int i = 0; i < EXP_SPT; ++i) { s_boff[tid * EXP_SPT + i] = eb; eb += nb[i]; }
    if (tid == EXP_THREADS - 1) s_boff[EXP_TILE] = eb;
    __syncthreads();
    if (ns == 0 || totb == 0) return;
    // (s_boff[ns .. EXP_TILE] all equal totb: the search and the walk stop at ns)
    u8 *out = bytes_out + tile_off[2 * tile];                     // first byte of the tile in the output stream
    const uintptr_t a0 = (uintptr_t)out;
    const u32 head = (u32)((8 - (a0 & 7)) & 7) < totb ? (u32)((8 - (a0 & 7)) & 7) : totb;       // bytes before the first aligned word
    const u32 nwords = (totb - head) >> 3;
    const u32 tail0 = head + nwords * 8;                          // first byte behind the last aligned word
    u64 *outw = reinterpret_cast<u64 *>(out + head);
    for (u32 w = tid; w < nwords; w += EXP_THREADS)
        outw[w] = pack_pull(s_boff, s_gpos, s_len, ns, head + 8 * w, 8, src8, src_bit0, src_words);
    if ((u32)tid < head) out[tid] = (u8)pack_pull(s_boff, s_gpos, s_len, ns, (u32)tid, 1, src8, src_bit0, src_words);
    if ((u32)tid < totb - tail0) out[tail0 + tid] = (u8)pack_pull(s_boff, s_gpos, s_len, ns, tail0 + tid, 1, src8, src_bit0, src_words);
}

} // namespace hsk
