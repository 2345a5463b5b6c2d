// hsk_combine.h -- the combining extraction: k-mers are counted where they are born, per MINIMIZER BUCKET, and only the distinct
// {k-mer, count} pairs of a bucket enter the radix pass (one GPU, one-word keys, no payload; from 64 MB of packed reads on).
//
// The reference extracts every k-mer instance of a task, sorts all of them and counts runs (src/kmerops.cpp:1382-1445); so does the
// instance path of this library (hsk_scatter.h + hsk_sort.h + hsk_agg.h), which writes every instance to HBM once and moves it once
// before the LDS tables of the finish see it: 4 x 8 bytes of HBM traffic per instance.  Sequencing data repeats every k-mer
// ~coverage times, and all instances of a canonical k-mer share its minimizer: the supermers that carry them can meet in one
// (task, minimizer) bucket long before anything is sorted -- and a supermer is 16 bytes for ~8 k-mers.  So:
//   1. scan_kernel hands out 32 mixed bits of every supermer's minimizer hash (ParseArgs::tile_sub), splits every task into 16
//      VIRTUAL tasks by the top four of them (ParseArgs::vt_shift: the parse's counting sort takes the first bits in its stride) and
//      cuts no supermer longer than 16 k-mers; place_items_kernel (hsk_parse.h) writes the supermer ITSELF to its slot -- a 16-byte
//      item: 64 bases + the k-mer count -- reading the packed reads once, in order;
//   2. bucket_hist / bucket_scan / bucket_scatter order the items of a virtual task by the next <= 10 minimizer bits: buckets of
//      ~12 k k-mers (8192 items staged and ordered in LDS per step, every bucket's run written in one piece);
//   3. combine_kernel: a workgroup takes a bucket, reads its items back to back, rolls their k-mers straight into a 2048-slot LDS
//      hash table (the probe loop of the finish, agg_count_keys, behind a first sweep that finds most k-mers in their home slots)
//      and then writes the table's {key, count} pairs into the chunk store of the first radix pass exactly as expand_scatter2_kernel
//      writes keys (cursor / map / chunk protocol of hsk_scatter.h), the counts into the payload chunks; a table that fills up in
//      the middle of a bucket is written out and started again, so a key may leave a bucket in several partial pairs: nothing
//      downstream assumes otherwise;
//   4. the second radix pass carries the counts as the payload (onesweep_multi_kernel<1, true>), and the finish adds them up instead
//      of counting records (agg_finish_kernel<cap, true>) over bins of the top 14 key bits: same order, same filter, same list.
// At 32 instances per k-mer the pass and the finish move 1/26 of the records (a k-mer next to a read end misses some windows); what
// remains per instance is the roll and one LDS read + add.  With (nearly) unique k-mers the pairs are as many as the instances and
// 16 instead of 8 bytes each: the host watches the ratio (more than one pair per sixteen k-mers: reads with errors, low coverage) and goes back to the instance path
// (hsk_ctx::combine_off); an item-mode store holds nothing the instance path could read, so a call that cannot go on here (a bin
// beyond the weighted finish's last table, no batch of eight tasks, a parse that left its fast path) starts again from the reads
// in HBM (HSK_RETRY_PLAN, dispatch_pipeline).
#pragma once
#include "hsk_parse.h"
#include "hsk_expand.h"
#include "hsk_scatter.h"
#include "hsk_agg.h"

namespace hsk {

// ---- 2. bucket order of the supermer items -------------------------------------------------------------------------------------
// The parse has already split every task by the top vt_shift minimizer bits ("virtual tasks", ParseArgs::vt_shift: the placement's
// counting sort takes 16 x as many bins in its stride); what is left is at most 10 bits per virtual task.  A workgroup stages 8192
// items in LDS, orders them by those bits there and writes every bucket's run (8 items = 128 bytes on average) in one piece.
constexpr int CS_THREADS = 1024;
constexpr int CS_IPT = 8;
constexpr u32 CS_TILE = CS_THREADS * CS_IPT;         // items staged per step
constexpr int CS_MAX_LOCAL = 10;                     // bucket bits inside a virtual task at most (one lane per bucket)
constexpr int CS_MAX_LOG2NB = 14;                    // buckets per task at most
constexpr u32 CS_ITEM = 1u << 17;                    // items per workgroup at most

struct BucketItem { u64 first; u32 n; u16 task; u16 hi; };   // supermer slots [first, first + n): task `task`, top minimizer bits `hi` (virtual task)
struct BucketSortArgs {
    const BucketItem *items;
    const u32 *sm_sub; const ulonglong2 *sm_item;
    u32 *off;                  // [ntasks][stride]: counts, then (bucket_scan_kernel) exclusive offsets with the total behind the last bucket
    u32 *cur;                  // [ntasks][stride]: running cursors of the scatter
    const u32 *log2nb;         // [ntasks] buckets of every task (log2)
    const u64 *out_base;       // [ntasks] first item of every task in `recs`
    u32 stride;
    u32 vt_shift;              // minimizer bits the virtual tasks have consumed
    u32 *err;                  // sticky error word (bit 64: an item list whose virtual task does not fit its task's buckets: nothing is written)
    ulonglong2 *recs;          // out: the items, tasks back to back, buckets ascending inside a task
    uint2 *units;              // out (bucket_units_kernel): the work units of the combining extraction, {first item, end} inside the task
    const u64 *unit_off;       // [ntasks] first unit of every task in `units`
    u32 *nunits;               // [ntasks] out: units of every task
};
// A work unit of the combining extraction is a bucket, or -- a minimizer that very many supermers share: homopolymers, satellites -- a slice of
// CB_UNIT items of one: the slices of a bucket go to different workgroups, every one counts into a table of its own and the finish adds
// their pairs up (as it does for a bucket whose table ran over).  Measured on 5 Gbp with 2 % of the reads replaced by all-A reads (one
// bucket of 10 M items, counted by ONE workgroup while the other 1279 had long finished): extraction 114 instead of 19 ms.
constexpr u32 CB_UNIT = 8192;
// bucket of an item inside its task = top lg bits of sub = {virtual task bits, local bits}
struct BucketMap { u32 lg, lgl, gbase; };
__device__ __forceinline__ BucketMap bucket_map(const BucketSortArgs &a, const BucketItem &it)
{
    BucketMap m; m.lg = a.log2nb[it.task];
    m.lgl = m.lg > a.vt_shift ? m.lg - a.vt_shift : 0u;
    m.gbase = m.lg >= a.vt_shift ? ((u32)it.hi << m.lgl) : ((u32)it.hi >> (a.vt_shift - m.lg));
    return m;
}
// a work item must address buckets of its own task only (anything else would be a bug upstream, never an address)
__device__ __forceinline__ bool bucket_map_ok(const BucketSortArgs &a, const BucketMap &m)
{
    const bool ok = m.lg <= (u32)CS_MAX_LOG2NB && m.lgl <= (u32)CS_MAX_LOCAL && (u64)m.gbase + (1ULL << m.lgl) <= (1ULL << m.lg);
    if (!ok && threadIdx.x == 0) atomicOr(a.err, 64u);
    return ok;
}
__device__ __forceinline__ u32 bucket_local(u32 sub, const BucketMap &m) { return m.lgl ? (sub >> (32 - m.lg)) & ((1u << m.lgl) - 1u) : 0u; }

// scan-placed bins (hsk_parse.h: bin_place) -> the bucket order's work list: one work item per chunk, in allocation order
struct BinItemsArgs { const u32 *cursor; const u32 *map; u32 vmax; const u32 *chunk_bin; u32 nchunks; u32 nvt; u32 vt_shift; BucketItem *items; };
__global__ __launch_bounds__(256) void bins_items_kernel(BinItemsArgs a)
{
    const u32 ph = blockIdx.x * 256u + threadIdx.x;
    if (ph >= a.nchunks) return;
    const u32 cb = a.chunk_bin[ph], bin = cb & 8191u, v = cb >> 13, vt = bin % a.nvt;      // (a chunk knows its bin and its place in the bin's list)
    const u64 cur = a.cursor[(u64)bin * BIN_CUR_STRIDE], lo = (u64)v * BIN_CHUNK;
    BucketItem it;
    it.first = (u64)ph * BIN_CHUNK;
    it.n = cur <= lo ? 0u : (cur - lo >= BIN_CHUNK ? BIN_CHUNK : (u32)(cur - lo));         // (a chunk opened ahead may have stayed empty)
    it.task = (u16)(vt >> a.vt_shift); it.hi = (u16)(vt & ((1u << a.vt_shift) - 1u));
    a.items[ph] = it;
}

__global__ __launch_bounds__(CS_THREADS) void bucket_hist_kernel(BucketSortArgs a)
{
    __shared__ u32 s_h[1 << CS_MAX_LOCAL];
    const BucketItem it = a.items[blockIdx.x];
    const BucketMap m = bucket_map(a, it);
    if (!bucket_map_ok(a, m)) return;
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const u32 *sub = a.sm_sub + it.first;
    for (u32 i0 = threadIdx.x; i0 < it.n; i0 += CS_THREADS * 4) {
        u32 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const u32 i = i0 + u * CS_THREADS; v[u] = i < it.n ? sub[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * CS_THREADS < it.n) atomicAdd(&s_h[bucket_local(v[u], m)], 1u);
    }
    __syncthreads();
    const u32 c = s_h[threadIdx.x];
    if (c) atomicAdd(&a.off[(u64)it.task * a.stride + m.gbase + threadIdx.x], c);
}

// one workgroup per task: counts -> exclusive offsets (in place, total behind the last bucket) and the scatter's cursors
__global__ __launch_bounds__(1024) void bucket_scan_kernel(BucketSortArgs a)
{
    __shared__ u32 s_w[16];
    const u32 t = blockIdx.x, nb = 1u << a.log2nb[t];
    u32 *g = a.off + (u64)t * a.stride, *cu = a.cur + (u64)t * a.stride;
    const u32 per = (nb + 1023u) / 1024u;
    const u32 lo = threadIdx.x * per;
    u32 sum = 0;
    for (u32 i = lo; i < lo + per && i < nb; ++i) sum += g[i];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan(sum);
    if (lane == WAVE - 1) s_w[w] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (int i = 0; i < 16; ++i) { const u32 s = s_w[i]; if (i < w) base += s; tot += s; }
    u32 run = base + inc - sum;
    for (u32 i = lo; i < lo + per && i < nb; ++i) { const u32 c = g[i]; g[i] = run; cu[i] = run; run += c; }
    if (threadIdx.x == 0) g[nb] = tot;
}

// one workgroup per task, after bucket_scan_kernel: the buckets' offsets -> the unit list (empty buckets make no unit)
__global__ __launch_bounds__(1024) void bucket_units_kernel(BucketSortArgs a)
{
    __shared__ u32 s_w[16];
    const u32 t = blockIdx.x, nb = 1u << a.log2nb[t];
    const u32 *g = a.off + (u64)t * a.stride;
    uint2 *un = a.units + a.unit_off[t];
    const u32 per = (nb + 1023u) / 1024u;
    const u32 lo = threadIdx.x * per;
    u32 sum = 0;
    for (u32 i = lo; i < lo + per && i < nb; ++i) sum += (g[i + 1] - g[i] + CB_UNIT - 1u) / CB_UNIT;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan(sum);
    if (lane == WAVE - 1) s_w[w] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (int i = 0; i < 16; ++i) { const u32 s = s_w[i]; if (i < w) base += s; tot += s; }
    u32 run = base + inc - sum;
    for (u32 i = lo; i < lo + per && i < nb; ++i)
        for (u32 r0 = g[i], r1 = g[i + 1]; r0 < r1; r0 += CB_UNIT) un[run++] = make_uint2(r0, r1 - r0 > CB_UNIT ? r0 + CB_UNIT : r1);
    if (threadIdx.x == 0) a.nunits[t] = tot;
}

__global__ __launch_bounds__(CS_THREADS) void bucket_scatter_kernel(BucketSortArgs a)
{
    __shared__ ulonglong2 s_it[CS_TILE];               // 128 KB: the step's items in bucket order
    __shared__ unsigned short s_bk[CS_TILE];           // their buckets
    __shared__ u32 s_cnt[1 << CS_MAX_LOCAL], s_start[1 << CS_MAX_LOCAL], s_gb[1 << CS_MAX_LOCAL];
    __shared__ u32 s_w[16];
    const int tid = threadIdx.x;
    const BucketItem it = a.items[blockIdx.x];
    const BucketMap m = bucket_map(a, it);
    if (!bucket_map_ok(a, m)) return;
    const u32 *sub = a.sm_sub + it.first;
    const ulonglong2 *src = a.sm_item + it.first;
    ulonglong2 *out = a.recs + a.out_base[it.task];
    u32 *cu = a.cur + (u64)it.task * a.stride + m.gbase;
    const int lane = lane_id(), w = tid >> 6;
    for (u32 t0 = 0; t0 < it.n; t0 += CS_TILE) {
        const u32 nt = it.n - t0 < CS_TILE ? it.n - t0 : CS_TILE;
        s_cnt[tid] = 0;
        __syncthreads();                                  // (also: the previous step's output loop is done with the stage)
        ulonglong2 x[CS_IPT]; u32 bk[CS_IPT], rk[CS_IPT];
#pragma unroll
        for (int u = 0; u < CS_IPT; ++u) {
            const u32 i = (u32)u * CS_THREADS + (u32)tid;
            const bool ok = i < nt;
            x[u] = ok ? src[t0 + i] : make_ulonglong2(0, 0);
            bk[u] = ok ? bucket_local(sub[t0 + i], m) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < CS_IPT; ++u) rk[u] = bk[u] != 0xFFFFFFFFu ? atomicAdd(&s_cnt[bk[u]], 1u) : 0u;
        __syncthreads();
        {
            const u32 c = s_cnt[tid];
            const u32 inc = wave_incl_scan(c);
            if (lane == WAVE - 1) s_w[w] = inc;
            __syncthreads();
            u32 base = 0;
            for (int i = 0; i < 16; ++i) if (i < w) base += s_w[i];
            const u32 st = base + inc - c;
            s_start[tid] = st;
            s_gb[tid] = (c ? atomicAdd(&cu[tid], c) : 0u) - st;      // (mod 2^32: global slot = s_gb + position in the stage)
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CS_IPT; ++u) {
            if (bk[u] == 0xFFFFFFFFu) continue;
            const u32 p = s_start[bk[u]] + rk[u];
            s_it[p] = x[u]; s_bk[p] = (unsigned short)bk[u];
        }
        __syncthreads();
        for (u32 i = tid; i < nt; i += CS_THREADS) out[s_gb[s_bk[i]] + i] = s_it[i];
    }
}

// ---- 1b. several ranks: the items are built by the OWNER of a task (round 4) ------------------------------------------------------
// The supermers travel as byte runs (hsk_comm.h: len[], bytes[] and, when the combining extraction is planned, the top 16 minimizer
// bits sub16[] -- 2 bytes per supermer on the wire).  For a batch of eight owned tasks, whose segments come from every source rank:
//   vt_hist_kernel     counts the supermers per (task, top four minimizer bits): the 16 VIRTUAL tasks the one-GPU parse makes itself
//   vt_scan_kernel     lays the (task, virtual task) runs out back to back
//   items_build_kernel a supermer's byte offset from the tile offsets of expand_scan_kernel + a scan of the tile's lengths, its first
//                      64 bases as the 16-byte item place_items_kernel would have written, scattered into the 16 runs of its task (a
//                      tile of 512 supermers: runs of ~32 items = 512 bytes)
// What follows is the one-GPU path unchanged: bucket order inside the virtual tasks, combine_kernel, one radix pass, weighted finish.
struct ItemBuildTask { const ExpSeg *segs; int nseg; const u8 *sm_len; const unsigned short *sub16; const u64 *src8; u64 src_words; const u64 *tile_off; u64 ntiles; };
struct ItemBuildArgs {
    ItemBuildTask t[8];
    u32 *vt_cnt;               // [8][16] supermers per (task of the batch, virtual task)
    u64 *vt_cur;               // [8][16] running slot cursors (vt_scan_kernel: the runs' first slots)
    ulonglong2 *items; u32 *subs;
    int k; u32 *err;
};

// (16 counters under 512 LDS atomics per tile cost 1.4 ms per batch, measured: here a wave takes a tile, a lane eight consecutive supermers -- one
//  16-byte load -- and counts into one of 16 copies of the counters; a workgroup walks many tiles: the minimizer bits of a batch are 200 MB, read once)
__global__ __launch_bounds__(EXP_THREADS) void vt_hist_kernel(ItemBuildArgs a)
{
    __shared__ u32 s_h[EXP_THREADS / WAVE][16][16];
    const ItemBuildTask &t = a.t[blockIdx.y];
    if ((u64)blockIdx.x * (EXP_THREADS / WAVE) >= t.ntiles) return;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (EXP_THREADS / WAVE) * 256; i += EXP_THREADS) (&s_h[0][0][0])[i] = 0;
    __syncthreads();
    u32 *mine = &s_h[w][lane & 15][0];
    for (u64 tile = (u64)blockIdx.x * (EXP_THREADS / WAVE) + (u64)w; tile < t.ntiles; tile += (u64)gridDim.x * (EXP_THREADS / WAVE)) {
        const ExpSeg seg = t.segs[seg_of_tile(t.segs, t.nseg, tile)];
        const u64 first = (tile - seg.tile_start) * EXP_TILE + 8u * (u32)lane;
        if (first >= seg.n_sup) continue;
        const u32 n_ok = seg.n_sup - first < 8 ? (u32)(seg.n_sup - first) : 8u;
        const uint4 s8 = *reinterpret_cast<const uint4 *>(t.sub16 + seg.sup_off + first);      // (unaligned; behind the last supermer: padding or the next segment)
        const u32 sw[4] = {s8.x, s8.y, s8.z, s8.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) if ((u32)j < n_ok) atomicAdd(&mine[(sw[j >> 1] >> (16 * (j & 1) + 12)) & 15u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        u32 c = 0;
        for (int i = 0; i < (EXP_THREADS / WAVE) * 16; ++i) c += (&s_h[0][0][0])[i * 16 + threadIdx.x];
        if (c) atomicAdd(&a.vt_cnt[blockIdx.y * 16 + threadIdx.x], c);
    }
}

__global__ __launch_bounds__(128) void vt_scan_kernel(ItemBuildArgs a)
{
    __shared__ u32 s_w[2];
    const u32 c = a.vt_cnt[threadIdx.x];
    const u32 inc = wave_incl_scan(c);
    if (lane_id() == WAVE - 1) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    a.vt_cur[threadIdx.x] = (u64)((threadIdx.x >> 6) ? s_w[0] : 0u) + inc - c;      // (a batch's supermers: below 2^32, checked on the host)
}

// A workgroup takes IB_TILES tiles, a wave one of them: a lane holds EIGHT CONSECUTIVE supermers (their lengths: one 8-byte load, their minimizer
// bits: one 16-byte load; byte offsets from a wave scan -- no workgroup barrier), the tile's byte run is staged in LDS with 16-byte loads and the
// items are cut from there; then the items are staged in virtual-task order (same LDS) and leave as runs of ~128 items (2 KB) from consecutive
// lanes.  (A lane per supermer, 1- and 2-byte loads, eight workgroup scans and stores into 16 runs per wave instruction: 2.26 ms per batch of
// 104 M supermers, this form ... -- measured)
constexpr int IB_TILES = 4;
constexpr int IB_N = IB_TILES * EXP_TILE;
constexpr u32 IB_TILE_BYTES = EXP_TILE * 19;           // a supermer of an item's length (<= K + 15 <= 70 bases) has at most 18 bytes
constexpr u32 IB_STAGE_BYTES = IB_TILE_BYTES + 64;     // + the run's lead behind a 16-byte boundary + the reach of the last supermer's 20-byte window
constexpr u32 IB_STAGE_WORDS = IB_STAGE_BYTES / 4;
static_assert(EXP_TILE == 8 * WAVE && IB_TILES * WAVE == EXP_THREADS, "a wave per tile, eight supermers per lane");
static_assert(IB_TILES * IB_STAGE_BYTES >= IB_N * 16 && IB_STAGE_BYTES % 16 == 0, "the item stage reuses the byte stage");
__global__ __launch_bounds__(EXP_THREADS) void items_build_kernel(ItemBuildArgs a)
{
    __shared__ __attribute__((aligned(16))) u32 s_raw[IB_TILES * IB_STAGE_WORDS];
    __shared__ unsigned short s_sb[IB_N];
    __shared__ u32 s_h[16], s_pre[16];
    __shared__ u64 s_base[16];
    ulonglong2 *s_it = reinterpret_cast<ulonglong2 *>(s_raw);
    const ItemBuildTask &t = a.t[blockIdx.y];
    const u64 tile0 = (u64)blockIdx.x * IB_TILES;
    if (tile0 >= t.ntiles) return;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    if (threadIdx.x < 16) s_h[threadIdx.x] = 0;
    const u64 tile = tile0 + (u64)w;
    u32 len[8], sub[8], boff[8], rk[8], n_ok = 0, lead = 0;
    bool fits = true;
    u32 *stage = s_raw + (u32)w * IB_STAGE_WORDS;
    if (tile < t.ntiles) {                                   // (uniform in the wave)
        const ExpSeg seg = t.segs[seg_of_tile(t.segs, t.nseg, tile)];
        const u64 first = (tile - seg.tile_start) * EXP_TILE + 8u * (u32)lane;
        n_ok = first < seg.n_sup ? (seg.n_sup - first < 8 ? (u32)(seg.n_sup - first) : 8u) : 0u;
        u64 l8 = 0; uint4 s8 = make_uint4(0, 0, 0, 0);
        if (n_ok) {                                          // (unaligned vector loads; what lies behind a segment's last supermer is the buffers' padding or the next segment)
            l8 = *reinterpret_cast<const u64 *>(t.sm_len + seg.sup_off + first);
            s8 = *reinterpret_cast<const uint4 *>(t.sub16 + seg.sup_off + first);
        }
        const u32 sw[4] = {s8.x, s8.y, s8.z, s8.w};
        u32 sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            len[j] = (u32)j < n_ok ? (u32)(l8 >> (8 * j)) & 255u : 0u;
            sub[j] = (sw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            boff[j] = sum; sum += (len[j] + 3u) >> 2;
        }
        const u32 inc = wave_incl_scan(sum);
        const u32 tile_bytes = __shfl(inc, WAVE - 1);
        const u32 excl = inc - sum;
        const u64 byte0 = t.tile_off[2 * tile];
        const u64 a16 = byte0 & ~15ULL;
        lead = (u32)(byte0 - a16) + excl;
        const u32 need = (u32)(byte0 - a16) + tile_bytes + 24u;
        fits = need <= IB_STAGE_BYTES;
        if (fits) {
            const u64 nbytes = t.src_words * 8;
            for (u32 o = 16u * (u32)lane; o < need; o += 16u * WAVE) {
                const u64 at = a16 + o;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (at + 16 <= nbytes) v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const u8 *>(t.src8) + at);
                else if (at + 8 <= nbytes) { const u64 x0 = t.src8[at >> 3]; v.x = (u32)x0; v.y = (u32)(x0 >> 32); }
                *reinterpret_cast<uint4 *>(stage + (o >> 2)) = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
            }
        } else if (lane == 0) atomicOr(a.err, 4u);         // (lengths no item has: the call fails)
    }
    __syncthreads();
    ulonglong2 it[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        it[j] = make_ulonglong2(0, 0); rk[j] = 0;
        if ((u32)j >= n_ok) continue;
        const u32 cnt = len[j] - (u32)a.k + 1u;
        if (len[j] < (u32)a.k || cnt > 16u) atomicOr(a.err, 4u);      // (the call fails; the slot is written all the same: the runs stay whole)
        else if (fits) {
            const u32 bit = 8u * (lead + boff[j]), i = bit >> 5, sh = bit & 31u;
            const u64 h0 = ((u64)stage[i] << 32) | stage[i + 1], h1 = ((u64)stage[i + 2] << 32) | stage[i + 3], h2 = (u64)stage[i + 4] << 32;
            const u64 w0 = sh ? ((h0 << sh) | (h1 >> (64 - sh))) : h0, w1 = sh ? ((h1 << sh) | (h2 >> (64 - sh))) : h1;
            it[j] = make_ulonglong2(w0, (w1 & ~0xFFULL) | (u64)cnt);
        }
        rk[j] = atomicAdd(&s_h[sub[j] >> 12], 1u);
    }
    __syncthreads();                                         // (every wave is done with its byte stage: the items take its place)
    if (threadIdx.x < WAVE) {
        const u32 c = threadIdx.x < 16 ? s_h[threadIdx.x] : 0u;
        const u32 inc = wave_incl_scan(c);
        if (threadIdx.x < 16) {
            s_pre[threadIdx.x] = inc - c;
            s_base[threadIdx.x] = c ? atomicAdd((unsigned long long *)&a.vt_cur[blockIdx.y * 16 + threadIdx.x], (unsigned long long)c) : 0ULL;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if ((u32)j >= n_ok) continue;
        const u32 p = s_pre[sub[j] >> 12] + rk[j];
        s_it[p] = it[j]; s_sb[p] = (unsigned short)sub[j];
    }
    __syncthreads();
    const u32 total = s_pre[15] + s_h[15];
    for (u32 i = threadIdx.x; i < total; i += EXP_THREADS) {
        const u32 sb = s_sb[i], v = sb >> 12;
        const u64 slot = s_base[v] + (u64)(i - s_pre[v]);
        a.items[slot] = s_it[i];
        a.subs[slot] = sb << 16;
    }
}

// ---- 3. the combining extraction -------------------------------------------------------------------------------------------------
constexpr int CB_THREADS = 256;
constexpr int CB_WAVES = CB_THREADS / WAVE;
#ifndef CB_LOG2CAP
#define CB_LOG2CAP 11
#endif
constexpr int CB_CAP = 1 << CB_LOG2CAP;
constexpr int CB_PER = CB_CAP / CB_THREADS;
#ifndef CB_GROUP
#define CB_GROUP 8
#endif
static_assert(CB_THREADS == 256, "one lane per digit in the table dump");
static_assert(CB_CAP <= (XS_SPAN - 1) * XsCfg<1>::CHUNK, "a dump's reservation touches at most XS_SPAN chunks of a digit");

struct CombineTask {
    const ulonglong2 *recs;    // the task's items in bucket order (place_items_kernel's two words)
    const uint2 *units;        // work units {first item, end}: a bucket or a slice of a large one (bucket_units_kernel)
    const u32 *nunits;         // their number (device memory)
    u32 nb;                    // buckets (0: no task on this XCD)
    u32 vmax;
    u32 cap_chunks;            // chunks the pair stores hold; the one behind them takes what does not fit (error bit 512: the host runs the call again with stores
                               // sized for the k-mers -- they are sized for the pairs the call's sketch of the input promises, four times over)
    u64 *chunks, *vchunks;     // chunk stores of the keys and of the counts (same slots)
    u64 *cursor; u32 *map; u32 *ctl; u64 *ghist;       // as ScatterTask; ctl[0]: bucket ticket
};
struct CombineArgs { CombineTask t[8]; int k, shift0, bits0, shift1; u32 *err; };      // first-pass digit: bits0 (<= 8) bits at shift0, second-pass digit: 8 bits at shift1 (both >= 32)

template <int KT = 0>
__global__ __launch_bounds__(CB_THREADS) void combine_kernel(CombineArgs a)
{
    constexpr int CHUNK = XsCfg<1>::CHUNK;
    __shared__ __attribute__((aligned(16))) u64 s_key[CB_CAP];
    __shared__ u32 s_val[CB_CAP];
    __shared__ u32 s_cnt[256], s_hist[256];
    __shared__ uint4 s_dl[256];
    __shared__ u32 s_scr[CB_WAVES];
    __shared__ u32 s_flag[4];                          // [0] bucket ticket, [1 + round % 3] a lane ran out of probes in that round
    typedef __attribute__((address_space(1))) u32 G32;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const int tid = threadIdx.x;
    const u32 xcc = __builtin_amdgcn_s_getreg(XCC_ID_GETREG) & 7u;
    const CombineTask &t = a.t[xcc];
    if (t.nb == 0) return;
    const u32 nunits = *t.nunits;
    const int k = KT ? KT : a.k;
    const int low = 64 - 2 * k;
    const u64 lastmask = ~0ULL << low;
    const u32 sh0 = (u32)a.shift0 - 32u, sh1 = (u32)a.shift1 - 32u, dm0 = (1u << a.bits0) - 1u;
    const u32 key_lds = (u32)(uintptr_t)(LdsPtr)s_key, val_lds = (u32)(uintptr_t)(LdsPtr)s_val;
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < CB_PER; ++j) { s_key[j * CB_THREADS + tid] = AG_EMPTY; s_val[j * CB_THREADS + tid] = 0; }
    s_cnt[tid] = 0; s_hist[tid] = 0;
    if (tid == 0) { s_flag[1] = 0; s_flag[2] = 0; s_flag[3] = 0; }
    u32 round = 0;                                     // insert rounds of this workgroup (the same in every wave)

    // The table's pairs leave for the chunk store of the first radix pass: digits counted, one reservation per digit (cursor, chunk map:
    // the protocol of expand_scatter2_kernel), every pair straight to its slot, the table empty again.  Called by the whole workgroup
    // behind a barrier (all inserts done); s_cnt is zero on entry and on exit.
    auto dump = [&]() {
        u64 mk[CB_PER]; u32 mv[CB_PER];
#pragma unroll
        for (int j = 0; j < CB_PER; ++j) {
            mk[j] = s_key[j * CB_THREADS + tid]; mv[j] = s_val[j * CB_THREADS + tid];
            if (mk[j] != AG_EMPTY) atomicAdd(&s_cnt[((u32)(mk[j] >> 32) >> sh0) & dm0], 1u);
        }
        xs_barrier();
        const u32 c = s_cnt[tid];
        u32 tot;
        const u32 st = block_excl_scan_xs<CB_WAVES>(c, s_scr, &tot);
        if (tot) {
            s_cnt[tid] = st;                                        // from a count to the running cursor of the digit's range
            if (c) {
                const u64 p = __hip_atomic_fetch_add(&t.cursor[tid], (u64)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const u64 v0 = p / CHUNK;
                const u32 off0 = (u32)(p % CHUNK);
                const u32 nv = (off0 + c - 1) / CHUNK + 1;
                G32 *mp = (G32 *)(t.map + (u64)tid * t.vmax);
                u32 ph[XS_SPAN] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < XS_SPAN; ++q) {
                    if ((u32)q >= nv || (q == 0 && off0 != 0)) continue;
                    ph[q] = __hip_atomic_fetch_add(&t.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
                    if (ph[q] > t.cap_chunks) { ph[q] = t.cap_chunks + 1u; atomicOr(a.err, 512u); }
                    __hip_atomic_store(mp + v0 + q, ph[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (off0 != 0) {
                    u32 spins = 0;
                    while ((ph[0] = __hip_atomic_load(mp + v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                        if (++spins > XS_SPIN_LIMIT) { atomicOr(a.err, 2u); ph[0] = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const u32 split = st + ((u32)CHUNK - off0);
                s_dl[tid] = make_uint4(split, (ph[0] - 1) * (u32)CHUNK + off0 - st, ((ph[1] ? ph[1] : 1u) - 1) * (u32)CHUNK - split,
                                       ((ph[2] ? ph[2] : 1u) - 1) * (u32)CHUNK - (split + (u32)CHUNK));
            }
            xs_barrier();
#pragma unroll
            for (int j = 0; j < CB_PER; ++j) {
                if (mk[j] == AG_EMPTY) continue;
                const u32 hi = (u32)(mk[j] >> 32);
                const u32 d = (hi >> sh0) & dm0;
                const u32 i = atomicAdd(&s_cnt[d], 1u);
                const uint4 dl = s_dl[d];
                const u32 o = i + (i < dl.x ? dl.y : (i < dl.x + (u32)CHUNK ? dl.z : dl.w));   // (mod 2^32)
                t.chunks[o] = mk[j]; t.vchunks[o] = (u64)mv[j];
                atomicAdd(&s_hist[(hi >> sh1) & 255u], 1u);
                s_key[j * CB_THREADS + tid] = AG_EMPTY; s_val[j * CB_THREADS + tid] = 0;
            }
            xs_barrier();
            s_cnt[tid] = 0;
        }
        xs_barrier();
    };

    for (;;) {
        xs_barrier();                                                 // (table cleared / previous bucket dumped; the ticket word is free)
        if (tid == 0) s_flag[0] = __hip_atomic_fetch_add(&t.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        xs_barrier();
        const u32 b = s_flag[0];
        if (b >= nunits) break;
        const uint2 un = t.units[b];
        const u32 r0 = un.x, r1 = un.y;
        ulonglong2 nxt = make_ulonglong2(0, 0);
        if (r0 + (u32)tid < r1) nxt = t.recs[r0 + (u32)tid];
        for (u32 base = r0; base < r1; base += CB_THREADS) {
            // ---- this lane's item: a supermer of up to 16 k-mers (the next tile's is asked for before this one is worked on) ----
            const ulonglong2 itm = nxt;
            { const u32 s2 = base + CB_THREADS + (u32)tid; nxt = s2 < r1 ? t.recs[s2] : make_ulonglong2(0, 0); }
            const u64 win0 = itm.x, win1 = itm.y & ~0xFFULL;
            u32 cnt = (u32)(itm.y & 0xFFULL);
            if (cnt > 16u) { atomicOr(a.err, 4u); cnt = 0; }
            // ---- its k-mers into the table.  A key's home is an aligned PAIR of slots.  First sweep: the homes of CB_GROUP k-mers are read
            //      together (one 16-byte LDS read each, one latency for the group); a k-mer that finds itself there -- at c-fold coverage
            //      all but one in c, minus the few whose key was pushed past its home -- just adds one.  The others are remembered as a bit
            //      mask and take the probe loop afterwards, every lane its next one per round: two or three rounds per tile instead of
            //      one probe loop per k-mer.  A lane whose probes run out keeps its remaining bits, the table is written out behind the
            //      round's barrier and the lanes go on (every such round starts on an empty table: it ends) ----
            u32 mm = cnt ? (0xFFFFu >> (16u - cnt)) : 0u;
            {
                Mer<1> fw, rc;
                fw.w[0] = win0 & lastmask;
                rc = twin<1>(fw, k);
#pragma unroll 1
                for (int q0 = 0; q0 < 16; q0 += CB_GROUP) {
                    if (__ballot((u32)q0 < cnt) == 0) break;
                    u64 key[CB_GROUP]; u32 h[CB_GROUP]; ulonglong2 cur[CB_GROUP];
#pragma unroll
                    for (int q = 0; q < CB_GROUP; ++q) {
                        const int r = q0 + q;
                        if (r > 0) {
                            fw.w[0] = funnel_left(win0, win1, 2 * r) & lastmask;
                            const u64 nbase = (fw.w[0] >> low) & 3;
                            rc.w[0] = ((rc.w[0] >> 2) | ((3 - nbase) << 62)) & lastmask;
                        }
                        key[q] = rc.w[0] < fw.w[0] ? rc.w[0] : fw.w[0];
                        h[q] = agg_slot<CB_LOG2CAP>(key[q]) & ~1u;
                        cur[q] = *reinterpret_cast<const ulonglong2 *>(&s_key[h[q]]);
                    }
#pragma unroll
                    for (int q = 0; q < CB_GROUP; ++q) {
                        const u32 r = (u32)(q0 + q);
                        if (r < cnt) {
                            if (cur[q].x == key[q]) { atomicAdd(&s_val[h[q]], 1u); mm &= ~(1u << r); }
                            else if (cur[q].y == key[q]) { atomicAdd(&s_val[h[q] + 1u], 1u); mm &= ~(1u << r); }
                        }
                    }
                }
            }
            for (;;) {
                bool stuck = false;
                for (;;) {
                    const bool mine = mm != 0 && !stuck;
                    const u64 act = __ballot(mine);
                    if (act == 0) break;
                    const u32 r = mine ? (u32)__builtin_ctz(mm) : 0u;
                    Mer<1> fw, rc;
                    fw.w[0] = (r ? funnel_left(win0, win1, 2 * (int)r) : win0) & lastmask;
                    rc = twin<1>(fw, k);
                    const u64 key = rc.w[0] < fw.w[0] ? rc.w[0] : fw.w[0];
                    u32 h = agg_slot<CB_LOG2CAP>(key) & ~1u;
                    const u64 left = agg_count_keys<(u32)CB_CAP - 1u>(act, key_lds, val_lds, h, key, 1u);
                    if (mine) { if ((left >> lane) & 1ULL) stuck = true; else mm &= mm - 1u; }
                }
                // one barrier per round: the round's flag is one of three taken in turn; the next round's is cleared before the barrier (its
                // last readers read it two rounds ago, i.e. before they arrived at the previous round's barrier)
                const u32 fl = 1u + round % 3u;
                if (tid == 0) s_flag[1u + (round + 1u) % 3u] = 0;
                if (mm) s_flag[fl] = 1;
                ++round;
                xs_barrier();
                if (s_flag[fl] == 0) break;
                dump();
            }
        }
        dump();
    }
    {
        const u32 cv = s_hist[tid];
        if (cv) atomicAdd((unsigned long long *)&t.ghist[tid], (unsigned long long)cv);
    }
}

// ---- 3b. two-word keys (40 <= K <= 55; round 4) ---------------------------------------------------------------------------------------
// The same plan for k-mers of two words: an item is still the supermer's first 64 bases, so it carries at most 61 - K k-mers (ParseArgs::
// item_maxk: the last four bases give way to the count); the table holds {word 1, word 0, count} and is filled by the probe loop of the
// two-word finish (agg2_count_keys: a slot is claimed on word 1, word 0 published behind it); the pairs leave as 16-byte records {word 0,
// word 1} into the chunk store of the first radix pass (digits: the top bits of word 1) with their counts beside them.  No first sweep over
// the home slots here: a 16-byte key has no 32-byte LDS read to take a pair of slots with.
#ifndef CB2_LOG2CAP
#define CB2_LOG2CAP 11
#endif
#ifndef C2_GROUP_N
#define C2_GROUP_N 4
#endif
template <int KT = 0>
__global__ __launch_bounds__(CB_THREADS) void combine2_kernel(CombineArgs a)
{
    constexpr int CHUNK = XsCfg<2>::CHUNK;
    constexpr int CB_CAP = 1 << CB2_LOG2CAP, CB_PER = CB_CAP / CB_THREADS, CB_LOG2CAP_ = CB2_LOG2CAP;
    static_assert(CB_CAP <= (XS_SPAN - 1) * CHUNK, "a dump's reservation touches at most XS_SPAN chunks of a digit");
    __shared__ u64 s_k1[CB_CAP];
    __shared__ u64 s_k0[CB_CAP];
    __shared__ u32 s_val[CB_CAP];
    __shared__ u32 s_cnt[256], s_hist[256];
    __shared__ uint4 s_dl[256];
    __shared__ u32 s_scr[CB_WAVES];
    __shared__ u32 s_flag[4];
    typedef __attribute__((address_space(1))) u32 G32;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const int tid = threadIdx.x;
    const u32 xcc = __builtin_amdgcn_s_getreg(XCC_ID_GETREG) & 7u;
    const CombineTask &t = a.t[xcc];
    if (t.nb == 0) return;
    const u32 nunits = *t.nunits;
    const int k = KT ? KT : a.k;                       // 33 .. 59
    const u64 lastmask = ~0ULL << (128 - 2 * k);       // the bases of word 1
    const u32 sh0 = (u32)a.shift0 - 32u, sh1 = (u32)a.shift1 - 32u, dm0 = (1u << a.bits0) - 1u;
    const u32 k1_lds = (u32)(uintptr_t)(LdsPtr)s_k1, k0_lds = (u32)(uintptr_t)(LdsPtr)s_k0, val_lds = (u32)(uintptr_t)(LdsPtr)s_val;
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < CB_PER; ++j) { s_k1[j * CB_THREADS + tid] = AG_EMPTY; s_k0[j * CB_THREADS + tid] = AG_EMPTY; s_val[j * CB_THREADS + tid] = 0; }
    s_cnt[tid] = 0; s_hist[tid] = 0;
    if (tid == 0) { s_flag[1] = 0; s_flag[2] = 0; s_flag[3] = 0; }
    u32 round = 0;

    auto dump = [&]() {                                 // as in combine_kernel, records of two words
        u64 m1[CB_PER], m0[CB_PER]; u32 mv[CB_PER];
#pragma unroll
        for (int j = 0; j < CB_PER; ++j) {
            m1[j] = s_k1[j * CB_THREADS + tid]; m0[j] = s_k0[j * CB_THREADS + tid]; mv[j] = s_val[j * CB_THREADS + tid];
            if (m1[j] != AG_EMPTY) atomicAdd(&s_cnt[((u32)(m1[j] >> 32) >> sh0) & dm0], 1u);
        }
        xs_barrier();
        const u32 c = s_cnt[tid];
        u32 tot;
        const u32 st = block_excl_scan_xs<CB_WAVES>(c, s_scr, &tot);
        if (tot) {
            s_cnt[tid] = st;
            if (c) {
                const u64 p = __hip_atomic_fetch_add(&t.cursor[tid], (u64)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const u64 v0 = p / CHUNK;
                const u32 off0 = (u32)(p % CHUNK);
                const u32 nv = (off0 + c - 1) / CHUNK + 1;
                G32 *mp = (G32 *)(t.map + (u64)tid * t.vmax);
                u32 ph[XS_SPAN] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < XS_SPAN; ++q) {
                    if ((u32)q >= nv || (q == 0 && off0 != 0)) continue;
                    ph[q] = __hip_atomic_fetch_add(&t.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
                    if (ph[q] > t.cap_chunks) { ph[q] = t.cap_chunks + 1u; atomicOr(a.err, 512u); }
                    __hip_atomic_store(mp + v0 + q, ph[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (off0 != 0) {
                    u32 spins = 0;
                    while ((ph[0] = __hip_atomic_load(mp + v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                        if (++spins > XS_SPIN_LIMIT) { atomicOr(a.err, 2u); ph[0] = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const u32 split = st + ((u32)CHUNK - off0);
                s_dl[tid] = make_uint4(split, (ph[0] - 1) * (u32)CHUNK + off0 - st, ((ph[1] ? ph[1] : 1u) - 1) * (u32)CHUNK - split,
                                       ((ph[2] ? ph[2] : 1u) - 1) * (u32)CHUNK - (split + (u32)CHUNK));
            }
            xs_barrier();
#pragma unroll
            for (int j = 0; j < CB_PER; ++j) {
                if (m1[j] == AG_EMPTY) continue;
                const u32 hi = (u32)(m1[j] >> 32);
                const u32 d = (hi >> sh0) & dm0;
                const u32 i = atomicAdd(&s_cnt[d], 1u);
                const uint4 dl = s_dl[d];
                const u32 o = i + (i < dl.x ? dl.y : (i < dl.x + (u32)CHUNK ? dl.z : dl.w));   // (mod 2^32)
                reinterpret_cast<ulonglong2 *>(t.chunks)[o] = make_ulonglong2(m0[j], m1[j]); t.vchunks[o] = (u64)mv[j];
                atomicAdd(&s_hist[(hi >> sh1) & 255u], 1u);
                s_k1[j * CB_THREADS + tid] = AG_EMPTY; s_k0[j * CB_THREADS + tid] = AG_EMPTY; s_val[j * CB_THREADS + tid] = 0;
            }
            xs_barrier();
            s_cnt[tid] = 0;
        }
        xs_barrier();
    };

    for (;;) {
        xs_barrier();
        if (tid == 0) s_flag[0] = __hip_atomic_fetch_add(&t.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        xs_barrier();
        const u32 b = s_flag[0];
        if (b >= nunits) break;
        const uint2 un = t.units[b];
        const u32 r0 = un.x, r1 = un.y;
        ulonglong2 nxt = make_ulonglong2(0, 0);
        if (r0 + (u32)tid < r1) nxt = t.recs[r0 + (u32)tid];
        for (u32 base = r0; base < r1; base += CB_THREADS) {
            const ulonglong2 itm = nxt;
            { const u32 s2 = base + CB_THREADS + (u32)tid; nxt = s2 < r1 ? t.recs[s2] : make_ulonglong2(0, 0); }
            const u64 win0 = itm.x, win1 = itm.y & ~0xFFULL;
            u32 cnt = (u32)(itm.y & 0xFFULL);
            if (cnt > 16u || (int)cnt + k - 1 > 60) { atomicOr(a.err, 4u); cnt = 0; }
            u32 mm = cnt ? (0xFFFFu >> (16u - cnt)) : 0u;
            {
                // first sweep, as in combine_kernel: the home slots of C2_GROUP k-mers are read together (both words: one LDS latency for the group instead of
                // one probe loop of the whole wave per k-mer); a k-mer that finds itself at home -- at c-fold coverage all but one in c -- adds one.  A slot
                // that is claimed but not published yet (word 0 still ~0) simply does not match: that k-mer takes the probe loop below, which waits.
                constexpr int C2_GROUP = C2_GROUP_N;
                const int low1 = 128 - 2 * k;
                Mer<2> fw, rc;
                fw.w[0] = win0; fw.w[1] = win1 & lastmask;
                rc = twin<2>(fw, k);
#pragma unroll 1
                for (int q0 = 0; q0 < 16; q0 += C2_GROUP) {
                    if (__ballot((u32)q0 < cnt) == 0) break;
                    u64 kw0[C2_GROUP], kw1[C2_GROUP], c1[C2_GROUP], c0[C2_GROUP]; u32 hh[C2_GROUP];
#pragma unroll
                    for (int q = 0; q < C2_GROUP; ++q) {
                        const int r = q0 + q;
                        if (r > 0) {
                            fw.w[0] = funnel_left(win0, win1, 2 * r);
                            fw.w[1] = (win1 << (2 * r)) & lastmask;
                            const u64 nb = (fw.w[1] >> low1) & 3;                        // the base that entered: the strand's last, the twin's first
                            rc.w[1] = ((rc.w[1] >> 2) | (rc.w[0] << 62)) & lastmask;
                            rc.w[0] = (rc.w[0] >> 2) | ((3 - nb) << 62);
                        }
                        // the smaller strand without four selects on one condition (v_cndmask_b32 x 4 behind one compare issues at a quarter of the rate of
                        // and / xor: profiles/r04_valu_rates.txt, cmp32_cnd4): a mask and three full-rate operations per half word
                        const u64 sel = mer_less<2>(rc, fw) ? ~0ULL : 0ULL;
                        kw0[q] = fw.w[0] ^ ((fw.w[0] ^ rc.w[0]) & sel); kw1[q] = fw.w[1] ^ ((fw.w[1] ^ rc.w[1]) & sel);
                        const u64 m = kw0[q] ^ (kw1[q] >> 9) ^ (kw1[q] << 21);
                        hh[q] = (((u32)(m >> 32) ^ (u32)m) * 0x9E3779B1u) >> (32 - CB_LOG2CAP_);
                        c1[q] = s_k1[hh[q]]; c0[q] = s_k0[hh[q]];
                    }
#pragma unroll
                    for (int q = 0; q < C2_GROUP; ++q) {
                        const u32 r = (u32)(q0 + q);
                        if (r < cnt && c1[q] == kw1[q] && c0[q] == kw0[q]) { atomicAdd(&s_val[hh[q]], 1u); mm &= ~(1u << r); }
                    }
                }
            }
            for (;;) {
                bool stuck = false;
                for (;;) {
                    const bool mine = mm != 0 && !stuck;
                    const u64 act = __ballot(mine);
                    if (act == 0) break;
                    const u32 r = mine ? (u32)__builtin_ctz(mm) : 0u;
                    Mer<2> fw, rc;
                    fw.w[0] = r ? funnel_left(win0, win1, 2 * (int)r) : win0;
                    fw.w[1] = (win1 << (2 * r)) & lastmask;
                    rc = twin<2>(fw, k);
                    const u64 sel = mer_less<2>(rc, fw) ? ~0ULL : 0ULL;
                    const u64 w0 = fw.w[0] ^ ((fw.w[0] ^ rc.w[0]) & sel), w1 = fw.w[1] ^ ((fw.w[1] ^ rc.w[1]) & sel);
                    const u64 m = w0 ^ (w1 >> 9) ^ (w1 << 21);
                    const u32 x = (u32)(m >> 32) ^ (u32)m;
                    u32 tmo = 0, h = (x * 0x9E3779B1u) >> (32 - CB_LOG2CAP_);
                    const u64 left = agg2_count_keys<(u32)CB_CAP - 1u>(act, k1_lds, k0_lds, val_lds, h, w1, w0, tmo);
                    if (mine) { if (((left >> lane) & 1ULL) || tmo) stuck = true; else mm &= mm - 1u; }
                }
                const u32 fl = 1u + round % 3u;
                if (tid == 0) s_flag[1u + (round + 1u) % 3u] = 0;
                if (mm) s_flag[fl] = 1;
                ++round;
                xs_barrier();
                if (s_flag[fl] == 0) break;
                dump();
            }
        }
        dump();
    }
    {
        const u32 cv = s_hist[tid];
        if (cv) atomicAdd((unsigned long long *)&t.ghist[tid], (unsigned long long)cv);
    }
}

} // namespace hsk
