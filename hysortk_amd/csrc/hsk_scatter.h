// hsk_scatter.h -- expand fused with the first scatter pass (aggregating finish; one-word keys with or without the
// EXTENSION payload, two-word keys without).
//
// The two-pass prefix plan (hsk_host_sort.h) orders a task by its top 16 key bits: a first pass on the lower 8 of them,
// whose output order inside a digit is free, and a second pass on the upper 8.  The first pass needs nothing but the
// keys, so it runs where the keys are born: expand_scatter_kernel rolls the k-mers of a tile exactly like
// expand_kernel (hsk_expand.h; reference GatheredSupermer::receive_from_buffer_stage2 src/kmerops.cpp:484-521) and,
// instead of writing them in input order for a radix pass to read back, ranks them by digit in LDS and writes them
// straight into the digit's bin: two record sizes of HBM traffic per k-mer less (the key array written by the expand and
// read by the first pass is never materialised).
//
// The digit histogram of the keys is not known before they exist, so a bin is not a pre-sized range but a LIST OF CHUNKS
// of CHUNK keys (= one tile of the second pass: 4096 one-word, 2048 two-word keys).  cursor[d] counts the keys reserved
// for digit d; a flush takes its range [p, p + c) with one atomic add; virtual chunk v = p / CHUNK of digit d lives in
// physical chunk map[d][v], allocated (bump counter) by the one reservation that contains the chunk's first slot and
// published through the map; everybody else whose range touches the chunk polls the map entry.  The allocator publishes
// before it waits for anything, so the wait is bounded by one L2 round trip (and by XS_SPIN_LIMIT: error word,
// HSK_ERR_INTERNAL).  A task wastes less than one chunk per digit: the chunk store holds n / CHUNK + 257 chunks.
//
// One task per XCD (HW_REG_XCC_ID, like onesweep_multi_kernel): cursors, map and chunk counter of a task are only ever
// touched from one XCD, so their atomics execute in that XCD's L2 (workgroup-scope RMW, L1-bypassing polls), and the
// short runs that neighbouring reservations of a digit write into the same 128-byte line merge in that L2 before they
// go to HBM.  The host checks afterwards that the cursors add up to the task's k-mer count.
//
// The kernel is a chain of short dependent phases per tile (prologue, roll + rank, digit scan, reservation, permute, run
// stores) on two 512-thread workgroups per CU.  What keeps the chain short: tiles are claimed two blocks ahead, so the
// next tile's supermer lengths, positions and first windows are in flight while the current tile is worked on; every
// barrier waits for LDS only (xs_barrier), so neither those loads nor the reservation atomic nor the run stores are
// drained at a barrier; the reservation is resolved while the keys are permuted.
//
// chunk_tiles_kernel then lists the chunks in (digit, virtual chunk) order as the tiles of the second pass
// (SortArgs::tile_src): tile = {physical chunk, keys in it}.  A tile holds one first-pass digit, so the second pass may
// rank its keys in any order too.
#pragma once
#include "hsk_expand.h"
#include "hsk_sort.h"

namespace hsk {

constexpr int XS_THREADS = 512;
constexpr int XS_WAVES = XS_THREADS / WAVE;
constexpr int XS_TILE = EXP_TILE;                     // supermers per tile: one per thread (the tile lists are shared with expand_kernel)
constexpr int XS_MAXSEG = 64;                        // segments (source ranks) whose tiles are looked up in LDS; more: no prefetch
constexpr int XS_CLAIM = 4;                          // tiles per claim
constexpr int XS_SPAN = 3;                            // chunks one reservation can touch
constexpr u32 XS_SPIN_LIMIT = 1u << 22;
// per key width: k-mers per work item (one-word keys: 16, nearly every supermer is one item; two-word keys: 8, the keys of
// an item stay in 32 registers either way), keys per chunk (= one tile of the second pass, 32 KB)
template <int NW> struct XsCfg {
    static constexpr int RUN = NW == 1 ? 16 : 8;
    static constexpr int MAX_ITEMS = XS_TILE * (128 / RUN);
    static constexpr int CHUNK = SortTile<NW>::TILE;
    static_assert(XS_THREADS * RUN <= (XS_SPAN - 1) * CHUNK, "a reservation touches at most XS_SPAN chunks of a digit");
    static_assert(2 * (RUN - 1) < 32, "the k-mers of an item start inside the first 32 bits of its window (funnel_left)");
    static_assert((CHUNK & (CHUNK - 1)) == 0, "chunk size is a power of two");
};
static_assert(XS_THREADS == XS_TILE, "one supermer per thread in the tile prologue");

struct ScatterTask {
    const ExpSeg *segs; int nseg; u32 vmax;          // vmax: map entries per digit (n / CHUNK + 1)
    const u8 *sm_len; const u64 *src8; u64 src_bit0, src_words;
    const u64 *sm_gpos; const u64 *tile_off; u64 ntiles;
    const u32 *sm_boff;                              // byte-store mode: supermer s starts at byte seg.byte_off + sm_boff[s] of src8
    u64 *chunks;                                     // chunk store (records of NW words)
    u64 *cursor;                                     // [256] keys reserved per digit (zeroed)
    u32 *map;                                        // [256][vmax] physical chunk + 1 (zeroed)
    u32 *ctl;                                        // [0] tile ticket, [1] chunks handed out (zeroed)
    u64 *ghist;                                      // [256] histogram of the second pass's digit (zeroed)
    u64 *tile_src;                                   // out (chunk_tiles_kernel): second-pass tiles
    u64 n;                                           // k-mers of the task (chunk_tiles_kernel checks the cursors and the histogram against it)
    u64 *gbase;                                      // out (chunk_tiles_kernel): [256] exclusive scan of ghist = digit bases of the second pass
    u32 *ntiles_out;                                 // out (chunk_tiles_kernel): number of second-pass tiles
    const u32 *sm_pos; const int32_t *sm_rid;        // EXTENSION: position in read and read id of every supermer
    u64 *vchunks;                                    // EXTENSION: payload chunk store (same slots as `chunks`)
    u64 *n_out;                                      // optional out (chunk_tiles_kernel): records in the chunk store (combining extraction: n = ~0, not known before)
};
struct ScatterArgs { ScatterTask t[8]; int k, shift0, shift1, chunk; u32 *err; };     // shift0, shift1 >= 32 (the digits are in the top 16 bits)

// Workgroup barrier for LDS traffic only: waits for the wave's LDS operations, not for its outstanding global loads, stores
// and atomics (__syncthreads() drains those too, which would put every prefetch and the reservation round trip on the
// critical path).  Everything the waves of this kernel hand to each other between barriers goes through LDS.
__device__ __forceinline__ void xs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NWAVES>
__device__ __forceinline__ u32 block_excl_scan_xs(u32 v, u32 *scratch /* >= NWAVES */, u32 *total)
{
    const int lane = lane_id(), w = threadIdx.x >> 6;
    u32 inc = wave_incl_scan(v);
    if (lane == WAVE - 1) scratch[w] = inc;
    xs_barrier();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) { u32 s = scratch[i]; if (i < w) base += s; tot += s; }
    xs_barrier();
    if (total) *total = tot;
    return base + inc - v;
}

// bits [c, c + 64) of the 128-bit string a:b, 0 < c < 32
__device__ __forceinline__ u64 funnel_left(u64 a, u64 b, int c)
{
    const u32 ah = (u32)(a >> 32), al = (u32)a, bh = (u32)(b >> 32);
    return ((u64)__builtin_amdgcn_alignbit(ah, al, 32 - c) << 32) | (u64)__builtin_amdgcn_alignbit(al, bh, 32 - c);
}

// Diagnostic build only (-DHSK_DIAG): shader-clock sums per phase of a flush, stamped by thread 0 of every workgroup
#ifdef HSK_DIAG
__device__ unsigned long long g_xs_diag[16];
#define XS_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); xs_acc[i] += t_ - xs_acc[15]; xs_acc[15] = t_; } } while (0)
#else
#define XS_STAMP(i) do { } while (0)
#endif

// EXT: every k-mer carries pos | rid << 32 (reference include/kmer.hpp:350-360) = its item's base value + the round; the stage
// keeps (lane, round) per slot and the lanes' base values sit in LDS, so the payload is rebuilt when the run is written.
// KT: k as a compile-time constant with the two digits in the top 16 key bits (0: k and the digit shifts come from the
// arguments).  The reference fixes k at compile time (KMER_SIZE); here the default k gets its own instance.
// TPB: threads per workgroup = supermers per tile (512 everywhere; a 256-thread instance -- four workgroups per CU, half the flush --
// was measured in round 3 and is slower: DESIGN.md 6.3).
template <int NW, bool EXT, int KT = 0, int TPB = XS_THREADS>
__global__ __launch_bounds__(TPB, 4) void expand_scatter_kernel(ScatterArgs a)
{
    constexpr int XS_RUN = XsCfg<NW>::RUN, XS_CHUNK = XsCfg<NW>::CHUNK;
    constexpr int XS_T = TPB, XS_NWAVE = TPB / WAVE, XS_TL = TPB, XS_MAX_ITEMS = TPB * (128 / XS_RUN);
    constexpr int XS_STG = TPB == XS_THREADS ? XS_CHUNK : XS_CHUNK / 2;       // keys the LDS stage holds (one window of a flush)
    static_assert(TPB == XS_THREADS || (NW == 1 && !EXT), "the small-workgroup instance is built for one-word keys without payload");
    static_assert(TPB >= 256 && TPB % WAVE == 0, "256 digit lanes");
    static_assert(!(EXT && NW != 1), "the payload variant is built for one-word keys");
#ifdef HSK_DIAG
    __shared__ unsigned long long xs_acc[16];                 // (LDS: sixteen 64-bit accumulators in registers would cost the kernel its occupancy)
    if (threadIdx.x == 0) { for (int i = 0; i < 15; ++i) xs_acc[i] = 0; xs_acc[15] = __builtin_amdgcn_s_memtime(); }
#endif
    __shared__ u32 s_boff[XS_TL + 1];
    __shared__ u32 s_koff[XS_TL + 1];
    __shared__ u32 s_ioff[XS_TL + 1];
    __shared__ u16 s_isup[XS_MAX_ITEMS];
    __shared__ u64 s_gpos[XS_TL];
    __shared__ u64 s_stage[XS_STG * NW];
    __shared__ u64 s_vb[EXT ? XS_T : 1];
    __shared__ u16 s_src[EXT ? XS_STG : 1];
    __shared__ u32 s_cnt[256], s_start[256], s_hist[256];
    __shared__ uint4 s_dl[256];                                         // per digit {split, d0, d1, d2}: staged slot g goes to g + d0 (g < split), g + d1 (g < split + chunk), else g + d2
    __shared__ u32 s_scr[XS_NWAVE];
    __shared__ u32 s_blk[2];
    __shared__ u64 s_seg[4][XS_MAXSEG];                                 // {first supermer slot, supermers, first tile, first byte} of the task's segments
    typedef __attribute__((address_space(1))) u32 G32;
    const int tid = threadIdx.x;
    const u32 xcc = __builtin_amdgcn_s_getreg(XCC_ID_GETREG) & 7u;
    const ScatterTask &t = a.t[xcc];
    if (t.ntiles == 0) return;
    const int k = KT ? KT : a.k;
    const int low = 64 * NW - 2 * k;                       // unused low bits of the last word
    const u64 lastmask = ~0ULL << low;
    const u32 sh0 = KT ? 16u : (u32)a.shift0 - 32u, sh1 = KT ? 24u : (u32)a.shift1 - 32u;
    const int nseg = t.nseg;
    const bool single = nseg == 1;                         // one GPU: one segment, kept in scalar registers
    const bool segs_lds = nseg <= XS_MAXSEG;               // the next tile's inputs are prefetched
    const bool byboff = t.sm_boff != nullptr;
    const bool inplace = t.sm_gpos != nullptr || byboff;
    const u64 s0_sup = t.segs[0].sup_off, s0_n = t.segs[0].n_sup, s0_byte = t.segs[0].byte_off;
    if (!single && segs_lds && tid < nseg) { const ExpSeg sg = t.segs[tid]; s_seg[0][tid] = sg.sup_off; s_seg[1][tid] = sg.n_sup; s_seg[2][tid] = sg.tile_start; s_seg[3][tid] = sg.byte_off; }
    // first base of supermer `idx` (absolute slot) of a segment whose bytes start at seg_byte
    auto sup_pos = [&](u64 idx, u64 seg_byte) -> u64 { return byboff ? 4 * (seg_byte + (u64)t.sm_boff[idx]) : t.sm_gpos[idx]; };
    if (tid < 256) { s_cnt[tid] = 0; s_hist[tid] = 0; }

    // Tiles are claimed in blocks of XS_CLAIM, two blocks ahead: the tile that follows the current one is always known,
    // so its supermer lengths and positions (and then its first windows) are requested while the current tile is worked on.
    if (tid == 0) {
        s_blk[0] = __hip_atomic_fetch_add(&t.ctl[0], (u32)XS_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s_blk[1] = __hip_atomic_fetch_add(&t.ctl[0], (u32)XS_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    xs_barrier();
    u64 blk = s_blk[0], nblk = s_blk[1];
    u32 p_len = 0; u64 p_gpos = 0, p_raw[NW + 2], p_vb = 0; bool p_have = false, p_win = false;
    // lengths and positions of the tile's supermers (byte streams: the tile's input offset instead of the positions)
#pragma unroll
    for (int x = 0; x < NW + 2; ++x) p_raw[x] = 0;
    // a workgroup's tiles come in ascending order: the segment of a tile is found by moving a cursor forward (two cursors: the tile
    // being worked on and the one whose inputs are prefetched), not by searching the table from the start for every tile
    int sg_meta = 0, sg_tile = 0;
    auto prefetch_meta = [&](u64 tl) {
        p_have = segs_lds && tl < t.ntiles; p_len = 0; p_gpos = 0; p_vb = 0;
        if (p_have) {
            if (single) {
                const u64 sidx = tl * XS_TL + tid;
                if (sidx < s0_n) {
                    p_len = t.sm_len[s0_sup + sidx]; if (inplace) p_gpos = sup_pos(s0_sup + sidx, s0_byte);
                    if (EXT) p_vb = (u64)t.sm_pos[s0_sup + sidx] | ((u64)(u32)t.sm_rid[s0_sup + sidx] << 32);
                }
            } else {
                while (sg_meta + 1 < nseg && s_seg[2][sg_meta + 1] <= tl) ++sg_meta;
                const int sg = sg_meta;
                const u64 sidx = (tl - s_seg[2][sg]) * XS_TL + tid;
                if (sidx < s_seg[1][sg]) {
                    p_len = t.sm_len[s_seg[0][sg] + sidx]; if (inplace) p_gpos = sup_pos(s_seg[0][sg] + sidx, s_seg[3][sg]);
                    if (EXT) p_vb = (u64)t.sm_pos[s_seg[0][sg] + sidx] | ((u64)(u32)t.sm_rid[s_seg[0][sg] + sidx] << 32);
                }
            }
            if (!inplace) p_gpos = t.tile_off[2 * tl];
        }
    };
    auto prefetch_win = [&]() {                            // speculation: item `tid` of the tile is the start of supermer `tid`
        p_win = p_have && inplace;
        if (p_win) {
            const u64 wi = (t.src_bit0 + 2 * p_gpos) >> 6;
#pragma unroll
            for (int x = 0; x < NW + 2; ++x) p_raw[x] = (wi + x < t.src_words) ? t.src8[wi + x] : 0;
        }
    };
    prefetch_meta(blk); prefetch_win();
    xs_barrier();                                       // s_blk[1] is rewritten during the first tile

    for (u32 j = 0;;) {
        const u64 tile = blk + j;
        if (tile >= t.ntiles) break;                                  // uniform; blocks come in ascending order
        const u64 ntile = (j + 1 == (u32)XS_CLAIM) ? nblk : tile + 1;
        u32 claim = 0;
        if (j == 0 && tid == 0) claim = __hip_atomic_fetch_add(&t.ctl[0], (u32)XS_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        u64 sg_sup = s0_sup, sg_n = s0_n, sg_t0 = 0, sg_byte = s0_byte;
        if (single) { }
        else if (segs_lds) {
            while (sg_tile + 1 < nseg && s_seg[2][sg_tile + 1] <= tile) ++sg_tile;
            const int sg = sg_tile;
            sg_sup = s_seg[0][sg]; sg_n = s_seg[1][sg]; sg_t0 = s_seg[2][sg]; sg_byte = s_seg[3][sg];
        } else { const ExpSeg *sp = t.segs + seg_of_tile(t.segs, nseg, tile); sg_sup = sp->sup_off; sg_n = sp->n_sup; sg_t0 = sp->tile_start; sg_byte = sp->byte_off; }
        const u64 first = (tile - sg_t0) * XS_TL;
        const u32 ns = (u32)((sg_n - first) < (u64)XS_TL ? (sg_n - first) : (u64)XS_TL);

        // ---- prologue: thread s owns supermer s of the tile ---------------------------------------------------
        u32 len = p_len; u64 gp = p_gpos;
        const bool have_win = p_win, had_meta = p_have;
        u64 svb = p_vb;                                               // EXTENSION: base value of supermer `tid`
        if (EXT && !p_have && (u32)tid < ns) svb = (u64)t.sm_pos[sg_sup + first + tid] | ((u64)(u32)t.sm_rid[sg_sup + first + tid] << 32);
        u64 rawp[NW + 2];
#pragma unroll
        for (int x = 0; x < NW + 2; ++x) rawp[x] = p_raw[x];
        if (!p_have) {
            len = ((u32)tid < ns) ? t.sm_len[sg_sup + first + tid] : 0;
            gp = (inplace && (u32)tid < ns) ? sup_pos(sg_sup + first + tid, sg_byte) : 0;
        }
        XS_STAMP(12);
        const u32 nb = ((u32)tid < ns) ? ((len + 3) >> 2) : 0;
        u32 nk = ((u32)tid < ns) ? (len - k + 1) : 0;
        if (nk > 128u) { atomicOr(a.err, 4u); nk = 0; }               // not a supermer of this library (at most 128 k-mers, at least one): the item tables would overflow
        const u32 ni = (nk + XS_RUN - 1) / XS_RUN;
        u32 tot2;
        const u32 e2 = block_excl_scan_xs<XS_NWAVE>((nk << 13) | ni, s_scr, &tot2);   // sums: items <= 4096, k-mers <= 65536
        const u32 ek = e2 >> 13, ei = e2 & 8191u, toti = tot2 & 8191u;
        u32 eb = 0;
        if (!inplace) eb = block_excl_scan_xs<XS_NWAVE>(nb, s_scr, nullptr);
        prefetch_meta(ntile);                                         // (after the last use of the values it replaces)
        XS_STAMP(13);
        s_boff[tid] = eb; s_koff[tid] = ek; s_ioff[tid] = ei;
        if (inplace) s_gpos[tid] = gp;
        for (u32 q = 0; q < ni; ++q) s_isup[ei + q] = (u16)tid;
        if (tid == XS_T - 1) { s_boff[XS_TL] = eb + nb; s_koff[XS_TL] = ek + nk; s_ioff[XS_TL] = ei + ni; }
        const u64 byte_abs = inplace ? 0 : (had_meta ? gp : t.tile_off[2 * tile]);
        xs_barrier();
        XS_STAMP(14);

        u32 n_cnt = 0, n_sh = 0; u64 n_raw[NW + 2], n_vb = 0;
        auto fetch = [&](u32 item) {
            n_cnt = 0; n_sh = 0;
#pragma unroll
            for (int x = 0; x < NW + 2; ++x) n_raw[x] = 0;
            if (item < toti) {
                const u32 sidx = s_isup[item];
                const u32 i0 = (item - s_ioff[sidx]) * XS_RUN;
                const u32 nks = s_koff[sidx + 1] - s_koff[sidx];
                n_cnt = nks - i0 < (u32)XS_RUN ? nks - i0 : (u32)XS_RUN;
                const u64 bit = inplace ? (t.src_bit0 + 2 * (s_gpos[sidx] + (u64)i0)) : (8 * (byte_abs + s_boff[sidx]) + 2 * (u64)i0);
                const u64 wi = bit >> 6; n_sh = (u32)(bit & 63);
#pragma unroll
                for (int x = 0; x < NW + 2; ++x) n_raw[x] = (wi + x < t.src_words) ? t.src8[wi + x] : 0;
                if (EXT) { const u64 sabs = sg_sup + first + sidx; n_vb = (u64)(t.sm_pos[sabs] + i0) | ((u64)(u32)t.sm_rid[sabs] << 32); }
            }
        };
        if (have_win && (ei == (u32)tid || (u32)tid >= toti)) {         // the speculation held for this lane (or it has no item)
            n_cnt = (u32)tid < toti ? (nk < (u32)XS_RUN ? nk : (u32)XS_RUN) : 0;
            n_sh = (u32)((t.src_bit0 + 2 * gp) & 63);
#pragma unroll
            for (int x = 0; x < NW + 2; ++x) n_raw[x] = rawp[x];
            n_vb = svb;
        } else fetch(tid);
        bool win_sent = false;
        XS_STAMP(0);                                                  // tile claim + prologue
        for (u32 it0 = 0; it0 < toti; it0 += XS_T) {
            const u32 cnt = n_cnt;
            if (EXT) s_vb[tid] = n_vb;                                // (the previous flush has been written: its last barrier is behind us)
            u64 win[NW + 1];
            {
                u64 aw[NW + 2];
#pragma unroll
                for (int x = 0; x < NW + 2; ++x) aw[x] = __builtin_bswap64(n_raw[x]);
#pragma unroll
                for (int x = 0; x < NW + 1; ++x) win[x] = n_sh ? ((aw[x] << n_sh) | (aw[x + 1] >> (64 - n_sh))) : aw[x];
            }
            if (it0 + XS_T < toti) fetch(it0 + XS_T + tid);
            // ---- roll the item's k-mers; rank every key inside its digit (arrival order: the pass is not stable) ----
            u64 key[XS_RUN][NW]; u32 rk[XS_RUN];
            Mer<NW> fw, rc;
#pragma unroll
            for (int x = 0; x < NW; ++x) fw.w[x] = win[x];
            fw.w[NW - 1] &= lastmask;
            rc = twin<NW>(fw, k);
#pragma unroll
            for (int r = 0; r < XS_RUN; ++r) {
                if (r > 0) {
                    // one base further: the k-mer at bit 2r of the window (two funnel shifts per word, the window itself stays);
                    // twin right by 2 bits, complement of the entering base on top
#pragma unroll
                    for (int x = 0; x < NW; ++x) fw.w[x] = funnel_left(win[x], win[x + 1], 2 * r);
                    fw.w[NW - 1] &= lastmask;
                    const u64 nbase = (fw.w[NW - 1] >> low) & 3;
#pragma unroll
                    for (int x = NW - 1; x > 0; --x) rc.w[x] = (rc.w[x] >> 2) | (rc.w[x - 1] << 62);
                    rc.w[0] = (rc.w[0] >> 2) | ((3 - nbase) << 62);
                    rc.w[NW - 1] &= lastmask;
                }
#pragma unroll
                for (int x = 0; x < NW; ++x) key[r][x] = 0;
                rk[r] = 0;
                if ((u32)r < cnt) {
                    const bool use_rc = mer_less<NW>(rc, fw);
#pragma unroll
                    for (int x = 0; x < NW; ++x) key[r][x] = use_rc ? rc.w[x] : fw.w[x];
                    const u32 hi = (u32)(key[r][NW - 1] >> 32);          // the digits are in the top 16 bits of the most significant word
                    rk[r] = atomicAdd(&s_cnt[(hi >> sh0) & 255u], 1u);
                    atomicAdd(&s_hist[(hi >> sh1) & 255u], 1u);
                }
            }
            if (!win_sent) { prefetch_win(); win_sent = true; }       // the next tile's positions have arrived by now
            XS_STAMP(1);                                              // window + roll + rank
            xs_barrier();                                          // digit counts of the flush complete
            XS_STAMP(2);

            // ---- flush: reserve the digits' ranges, permute through LDS, write runs ---------------------------
            const u32 c = tid < 256 ? s_cnt[tid] : 0;
            u32 tot;
            const u32 st = block_excl_scan_xs<XS_NWAVE>(c, s_scr, &tot);
            u64 p = 0;
            if (tid < 256) {
                s_start[tid] = st; s_cnt[tid] = 0;
                if (c) p = __hip_atomic_fetch_add(&t.cursor[tid], (u64)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            XS_STAMP(3);                                              // scan + reservation issued
            xs_barrier();
            // the sorted order of the flush goes through the stage in windows of XS_CHUNK keys (one window unless the
            // tile's supermers are longer than usual); the reservation is resolved while the first window is staged
#pragma unroll
            for (int r = 0; r < XS_RUN; ++r) {
                if ((u32)r >= cnt) continue;
                const u32 pos = s_start[((u32)(key[r][NW - 1] >> 32) >> sh0) & 255u] + rk[r];
                rk[r] = pos;                                           // position in the sorted order of the flush
                if (pos < (u32)XS_STG) {
#pragma unroll
                    for (int x = 0; x < NW; ++x) s_stage[pos * NW + x] = key[r][x];
                    if (EXT) s_src[pos] = (u16)((tid << 4) | r);
                }
            }
            XS_STAMP(4);                                              // sync + permute
            if (tid < 256 && c) {
                const u64 v0 = p / XS_CHUNK;
                const u32 off0 = (u32)(p % XS_CHUNK);
                const u32 nv = (off0 + c - 1) / XS_CHUNK + 1;          // chunks touched
                G32 *mp = (G32 *)(t.map + (u64)tid * t.vmax);
                u32 ph[XS_SPAN] = {0, 0, 0};
                // the chunks whose first slot is mine are allocated and published before anything is waited for
#pragma unroll
                for (int j = 0; j < XS_SPAN; ++j) {
                    if ((u32)j >= nv || (j == 0 && off0 != 0)) continue;
                    ph[j] = __hip_atomic_fetch_add(&t.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
                    __hip_atomic_store(mp + v0 + j, ph[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (off0 != 0) {                                      // the chunk my range starts in was opened by another reservation
                    u32 spins = 0;
                    while ((ph[0] = __hip_atomic_load(mp + v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                        if (++spins > XS_SPIN_LIMIT) { atomicOr(a.err, 2u); ph[0] = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const u32 split = st + ((u32)XS_CHUNK - off0);         // first slot (in the sorted order of the flush) in the second chunk
                s_dl[tid] = make_uint4(split, (ph[0] - 1) * (u32)XS_CHUNK + off0 - st, ((ph[1] ? ph[1] : 1u) - 1) * (u32)XS_CHUNK - split,
                                       ((ph[2] ? ph[2] : 1u) - 1) * (u32)XS_CHUNK - (split + (u32)XS_CHUNK));
            }
            XS_STAMP(5);                                              // reservation returned, chunk resolved
            for (u32 w0 = 0; w0 < tot; w0 += XS_STG) {
                if (w0) {
#pragma unroll
                    for (int r = 0; r < XS_RUN; ++r)
                        if ((u32)r < cnt && rk[r] >= w0 && rk[r] < w0 + (u32)XS_STG) {
#pragma unroll
                            for (int x = 0; x < NW; ++x) s_stage[(rk[r] - w0) * NW + x] = key[r][x];
                            if (EXT) s_src[rk[r] - w0] = (u16)((tid << 4) | r);
                        }
                }
                xs_barrier();
                const u32 wn = tot - w0 < (u32)XS_STG ? tot - w0 : (u32)XS_STG;
                for (u32 i = tid; i < wn; i += XS_T) {
                    u64 kw[NW];
#pragma unroll
                    for (int x = 0; x < NW; ++x) kw[x] = s_stage[i * NW + x];
                    const u32 d = ((u32)(kw[NW - 1] >> 32) >> sh0) & 255u;
                    const u32 g = w0 + i;
                    const uint4 dl = s_dl[d];
                    const u32 o = g + (g < dl.x ? dl.y : (g < dl.x + (u32)XS_CHUNK ? dl.z : dl.w));   // (mod 2^32)
                    if (NW == 2) *reinterpret_cast<ulonglong2 *>(t.chunks + (u64)o * 2) = make_ulonglong2(kw[0], kw[NW - 1]);
                    else {
#pragma unroll
                        for (int x = 0; x < NW; ++x) t.chunks[(u64)o * NW + x] = kw[x];
                    }
                    if (EXT) { const u32 src = s_src[i]; t.vchunks[o] = s_vb[src >> 4] + (u64)(src & 15u); }
                }
                xs_barrier();                                      // the stage is rewritten by the next window / flush
            }
            XS_STAMP(6);
#ifdef HSK_DIAG
            if (tid == 0) { xs_acc[10] += 1; xs_acc[11] += tot; }
#endif
        }
        if (!win_sent) prefetch_win();                                // (tile without k-mers)
        if (j == 0 && tid == 0) s_blk[1] = claim;                     // read at the next block switch, at least one barrier from here
        if (++j == (u32)XS_CLAIM) { xs_barrier(); blk = nblk; nblk = s_blk[1]; j = 0; xs_barrier(); }
    }
    if (tid < 256) {
        const u32 cv = s_hist[tid];
        if (cv) atomicAdd((unsigned long long *)&t.ghist[tid], (unsigned long long)cv);
    }
#ifdef HSK_DIAG
    if (tid == 0) for (int i = 0; i < 15; ++i) atomicAdd(&g_xs_diag[i], xs_acc[i]);
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// Two-sweep variant (one-word keys, no payload, bases read in place -- positions in the packed reads on one GPU, offsets into the
// received byte streams on several; HSK_XS2=0 takes expand_scatter_kernel instead, hsk_host_scatter.h).  expand_scatter_kernel keeps the
// 16 keys of a lane's item and their 16 ranks in registers from the roll to the permute: 48 of its 119 VGPRs, four waves per SIMD,
// two workgroups per CU -- and no unit of the CU more than 55 % busy (VALU 54 %, LDS 34 %: the chain of phases waits for itself).
// Here a flush rolls its k-mers TWICE: the first sweep only counts digits, the second -- after the digit scan, with the counts
// turned into running cursors -- rolls them again and drops every key straight into its slot of the stage.  Nothing survives
// between the sweeps but the item's two window words; the permute phase and its table are gone.  About 12 more VALU instructions
// per k-mer for 40 fewer registers and 10 KB less LDS: a third workgroup per CU.
// A flush with more keys than the stage holds (long supermers) is redone as two flushes over the two halves of the items.
template <class F>
__device__ __forceinline__ void xs2_roll(u64 win0, u64 win1, u32 cnt, int k, int low, u64 lastmask, F &&emit)
{
    Mer<1> fw, rc;
    fw.w[0] = win0 & lastmask;
    rc = twin<1>(fw, k);
#ifndef XS2_UNROLL
#define XS2_UNROLL 4
#endif
#pragma unroll XS2_UNROLL
    for (int r = 0; r < 16; ++r) {
        if (r > 0) {
            fw.w[0] = funnel_left(win0, win1, 2 * r) & lastmask;
            const u64 nbase = (fw.w[0] >> low) & 3;
            rc.w[0] = ((rc.w[0] >> 2) | ((3 - nbase) << 62)) & lastmask;
        }
        if ((u32)r < cnt) emit(rc.w[0] < fw.w[0] ? rc.w[0] : fw.w[0]);
    }
}

#ifndef XS2_WAVES
#define XS2_WAVES 6                                                      // waves per SIMD the kernel is built for: 6 = three workgroups per CU
#endif
template <int KT = 0>
__global__ __launch_bounds__(XS_THREADS, XS2_WAVES) void expand_scatter2_kernel(ScatterArgs a)
{
    constexpr int XS_RUN = 16, XS_CHUNK = XsCfg<1>::CHUNK, XS_STG = XS_CHUNK;
    __shared__ u64 s_stage[XS_STG];
    // tiles with supermers of more than 16 k-mers (rare) look their items up in tables that live in the stage between two flushes:
    // first item, k-mers and position of every supermer
    u16 *const s_ioff = reinterpret_cast<u16 *>(s_stage);               // [XS_TILE + 1]: bytes 0 .. 1026
    u8 *const s_nk = reinterpret_cast<u8 *>(s_stage) + 1536;            // [XS_TILE]:     bytes 1536 .. 2048
    u64 *const s_gpos = s_stage + 256;                                  // [XS_TILE]:     bytes 2048 .. 6144
    static_assert(XS_STG * 8 >= 2048 + XS_TILE * 8, "the item tables fit the stage");
    __shared__ u32 s_cnt[256], s_hist[256];
    __shared__ uint4 s_dl[256];
    __shared__ u32 s_scr[XS_WAVES];
    __shared__ u32 s_blk[2];
    __shared__ u32 s_multi;
    __shared__ u64 s_seg[4][XS_MAXSEG];                                 // {first supermer slot, supermers, first tile, first byte} of the task's segments
    typedef __attribute__((address_space(1))) u32 G32;
    const int tid = threadIdx.x;
    const u32 xcc = __builtin_amdgcn_s_getreg(XCC_ID_GETREG) & 7u;
    const ScatterTask &t = a.t[xcc];
    if (t.ntiles == 0) return;
    const int k = KT ? KT : a.k;
    const int low = 64 - 2 * k;
    const u64 lastmask = ~0ULL << low;
    const u32 sh0 = KT ? 16u : (u32)a.shift0 - 32u, sh1 = KT ? 24u : (u32)a.shift1 - 32u;
    const int nseg = t.nseg;
    const bool single = nseg == 1;
    const bool byboff = t.sm_boff != nullptr;              // byte-store mode: supermer s starts at byte seg.byte_off + sm_boff[s] of the stream
    const u64 s0_sup = t.segs[0].sup_off, s0_n = t.segs[0].n_sup, s0_byte = t.segs[0].byte_off;
    if (!single && tid < nseg) { const ExpSeg sg = t.segs[tid]; s_seg[0][tid] = sg.sup_off; s_seg[1][tid] = sg.n_sup; s_seg[2][tid] = sg.tile_start; s_seg[3][tid] = sg.byte_off; }
    if (tid < 256) { s_cnt[tid] = 0; s_hist[tid] = 0; }
    if (tid == 0) {
        s_multi = 0;
        s_blk[0] = __hip_atomic_fetch_add(&t.ctl[0], (u32)XS_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s_blk[1] = __hip_atomic_fetch_add(&t.ctl[0], (u32)XS_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    xs_barrier();
    u64 blk = s_blk[0], nblk = s_blk[1];
    u32 p_len = 0; u64 p_gpos = 0, p_r0 = 0, p_r1 = 0, p_r2 = 0; bool p_have = false;
    int sg_meta = 0, sg_tile = 0;
    auto prefetch_meta = [&](u64 tl) {
        p_have = tl < t.ntiles; p_len = 0; p_gpos = 0;
        if (p_have) {
            u64 base = s0_sup, n = s0_n, t0 = 0, byte0 = s0_byte;
            if (!single) {
                while (sg_meta + 1 < nseg && s_seg[2][sg_meta + 1] <= tl) ++sg_meta;
                base = s_seg[0][sg_meta]; n = s_seg[1][sg_meta]; t0 = s_seg[2][sg_meta]; byte0 = s_seg[3][sg_meta];
            }
            const u64 sidx = (tl - t0) * XS_TILE + tid;
            if (sidx < n) { p_len = t.sm_len[base + sidx]; p_gpos = byboff ? 4 * (byte0 + (u64)t.sm_boff[base + sidx]) : t.sm_gpos[base + sidx]; }
        }
    };
    auto load_win = [&](u64 gpos, u64 &r0, u64 &r1, u64 &r2) {
        const u64 wi = (t.src_bit0 + 2 * gpos) >> 6;
        r0 = (wi < t.src_words) ? t.src8[wi] : 0; r1 = (wi + 1 < t.src_words) ? t.src8[wi + 1] : 0; r2 = (wi + 2 < t.src_words) ? t.src8[wi + 2] : 0;
    };
    prefetch_meta(blk);
    if (p_have) load_win(p_gpos, p_r0, p_r1, p_r2);
    xs_barrier();

    for (u32 j = 0;;) {
        const u64 tile = blk + j;
        if (tile >= t.ntiles) break;
        const u64 ntile = (j + 1 == (u32)XS_CLAIM) ? nblk : tile + 1;
        u32 claim = 0;
        if (j == 0 && tid == 0) claim = __hip_atomic_fetch_add(&t.ctl[0], (u32)XS_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        u64 sg_n = s0_n, sg_t0 = 0;
        if (!single) { while (sg_tile + 1 < nseg && s_seg[2][sg_tile + 1] <= tile) ++sg_tile; sg_n = s_seg[1][sg_tile]; sg_t0 = s_seg[2][sg_tile]; }
        const u64 first = (tile - sg_t0) * XS_TILE;
        const u32 ns = (u32)((sg_n - first) < (u64)XS_TILE ? (sg_n - first) : (u64)XS_TILE);

        // ---- prologue: thread s owns supermer s of the tile (its length, position and first window were asked for a tile ago) ----
        const u32 len = p_len; const u64 gp = p_gpos;
        u64 raw0 = p_r0, raw1 = p_r1, raw2 = p_r2;
        u32 nk = ((u32)tid < ns) ? (len - k + 1) : 0;
        if (nk > 128u) { atomicOr(a.err, 4u); nk = 0; }
        const u32 ni = (nk + XS_RUN - 1) / XS_RUN;
        if (__ballot(ni > 1) != 0 && lane_id() == 0) s_multi = 1;       // a supermer of more than 16 k-mers: items and supermers part ways
        prefetch_meta(ntile);
        xs_barrier();
        const bool multi = s_multi != 0;
        u32 toti = ns, ei = (u32)tid;
        if (multi) {                                                   // (rare: repeats, homopolymers)
            ei = block_excl_scan_xs<XS_WAVES>(ni, s_scr, &toti);
            if (tid == 0) s_multi = 0;
        }
        bool win_sent = false;
        for (u32 it0 = 0; it0 < toti; it0 += XS_THREADS) {
            // ---- this lane's item: up to 16 k-mers of one supermer ----
            u32 cnt = 0; u64 ipos = gp;
            if (!multi) cnt = nk < (u32)XS_RUN ? nk : (u32)XS_RUN;       // item = supermer, window already here
            else {
                // the tables, written into the stage (the previous flush is behind its last barrier; the next write to the stage
                // is this flush's second sweep, two barriers from here)
                s_ioff[tid] = (u16)ei; s_nk[tid] = (u8)nk; s_gpos[tid] = gp;
                if (tid == XS_THREADS - 1) s_ioff[XS_TILE] = (u16)(ei + ni);
                xs_barrier();
                const u32 item = it0 + (u32)tid;
                if (item < toti) {
                    u32 lo = 0, hi = XS_TILE;                          // last supermer whose first item is <= item
                    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if ((u32)s_ioff[mid] <= item) lo = mid; else hi = mid; }
                    // (a supermer without k-mers shares its offset with its successor and is never the last one at or below `item`)
                    const u32 i0 = (item - (u32)s_ioff[lo]) * XS_RUN, nks = s_nk[lo];
                    cnt = nks > i0 ? (nks - i0 < (u32)XS_RUN ? nks - i0 : (u32)XS_RUN) : 0;
                    ipos = s_gpos[lo] + i0;
                    load_win(ipos, raw0, raw1, raw2);
                }
            }
            const u32 n_sh = (u32)((t.src_bit0 + 2 * ipos) & 63);
            u64 win0, win1;
            {
                const u64 a0 = __builtin_bswap64(raw0), a1 = __builtin_bswap64(raw1), a2 = __builtin_bswap64(raw2);
                win0 = n_sh ? ((a0 << n_sh) | (a1 >> (64 - n_sh))) : a0;
                win1 = n_sh ? ((a1 << n_sh) | (a2 >> (64 - n_sh))) : a1;
            }
            int npass = 1;
            for (int pass = 0; pass < npass; ++pass) {
                const u32 mycnt = (npass == 1 || (u32)(tid >> 8) == (u32)pass) ? cnt : 0u;
                // ---- sweep 1: digit counts ----
                xs2_roll(win0, win1, mycnt, k, low, lastmask, [&](u64 key) { atomicAdd(&s_cnt[((u32)(key >> 32) >> sh0) & 255u], 1u); });
                if (!win_sent) { if (p_have) load_win(p_gpos, p_r0, p_r1, p_r2); win_sent = true; }     // the next tile's positions have arrived by now
                xs_barrier();
                const u32 c = tid < 256 ? s_cnt[tid] : 0;
                u32 tot;
                const u32 st = block_excl_scan_xs<XS_WAVES>(c, s_scr, &tot);
                if (npass == 1 && tot > (u32)XS_STG) {                 // more keys than the stage holds: the two halves of the items one after the other
                    if (tid < 256) s_cnt[tid] = 0;
                    xs_barrier();
                    npass = 2; pass = -1;
                    continue;
                }
                u64 p = 0;
                if (tid < 256) {
                    s_cnt[tid] = st;                                   // from a count to the running cursor of the digit's range in the stage
                    if (c) p = __hip_atomic_fetch_add(&t.cursor[tid], (u64)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                xs_barrier();
                // ---- sweep 2: the k-mers again, each straight into its slot ----
                xs2_roll(win0, win1, mycnt, k, low, lastmask, [&](u64 key) {
                    const u32 hi = (u32)(key >> 32);
                    const u32 pos = atomicAdd(&s_cnt[(hi >> sh0) & 255u], 1u);
                    s_stage[pos] = key;
                    atomicAdd(&s_hist[(hi >> sh1) & 255u], 1u);
                });
                if (tid < 256 && c) {
                    const u64 v0 = p / XS_CHUNK;
                    const u32 off0 = (u32)(p % XS_CHUNK);
                    const u32 nv = (off0 + c - 1) / XS_CHUNK + 1;
                    G32 *mp = (G32 *)(t.map + (u64)tid * t.vmax);
                    u32 ph[XS_SPAN] = {0, 0, 0};
#pragma unroll
                    for (int q = 0; q < XS_SPAN; ++q) {
                        if ((u32)q >= nv || (q == 0 && off0 != 0)) continue;
                        ph[q] = __hip_atomic_fetch_add(&t.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
                        __hip_atomic_store(mp + v0 + q, ph[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (off0 != 0) {
                        u32 spins = 0;
                        while ((ph[0] = __hip_atomic_load(mp + v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                            if (++spins > XS_SPIN_LIMIT) { atomicOr(a.err, 2u); ph[0] = 1; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    const u32 split = st + ((u32)XS_CHUNK - off0);
                    s_dl[tid] = make_uint4(split, (ph[0] - 1) * (u32)XS_CHUNK + off0 - st, ((ph[1] ? ph[1] : 1u) - 1) * (u32)XS_CHUNK - split,
                                           ((ph[2] ? ph[2] : 1u) - 1) * (u32)XS_CHUNK - (split + (u32)XS_CHUNK));
                }
                xs_barrier();                                          // stage and digit table complete
                for (u32 i = tid; i < tot; i += XS_THREADS) {
                    const u64 kw = s_stage[i];
                    const u32 d = ((u32)(kw >> 32) >> sh0) & 255u;
                    const uint4 dl = s_dl[d];
                    const u32 o = i + (i < dl.x ? dl.y : (i < dl.x + (u32)XS_CHUNK ? dl.z : dl.w));   // (mod 2^32)
                    t.chunks[o] = kw;
                }
                if (tid < 256) s_cnt[tid] = 0;
                xs_barrier();                                          // the stage is rewritten by the next flush
            }
        }
        if (!win_sent && p_have) load_win(p_gpos, p_r0, p_r1, p_r2);
        if (j == 0 && tid == 0) s_blk[1] = claim;
        if (++j == (u32)XS_CLAIM) { xs_barrier(); blk = nblk; nblk = s_blk[1]; j = 0; xs_barrier(); }
    }
    if (tid < 256) {
        const u32 cv = s_hist[tid];
        if (cv) atomicAdd((unsigned long long *)&t.ghist[tid], (unsigned long long)cv);
    }
}

// tile_src[i] = (physical chunk << 32) | keys, chunks in (digit, virtual chunk) order; one workgroup per task.  The kernel
// also prepares everything else the second pass needs, so that the host does not have to read anything back between the
// two passes: the digit bases of the second pass (exclusive scan of the histogram the expand counted), the tile count, and
// the check that the XCD really expanded its task (cursors and histogram add up to the task's k-mer count; error bit 8
// otherwise: the kernels pick their task by the XCD they run on, the results must never silently depend on the placement).
__global__ __launch_bounds__(256) void chunk_tiles_kernel(ScatterArgs a)
{
    __shared__ u64 s_scr[8];
    const ScatterTask &t = a.t[blockIdx.x];
    if (t.ntiles == 0) { if (threadIdx.x == 0 && t.ntiles_out) *t.ntiles_out = 0; return; }
    const int d = threadIdx.x;
    const u64 cnt = t.cursor[d];
    const u64 CH = (u64)a.chunk;
    const u64 nch = (cnt + CH - 1) / CH;
    u64 tot_ch, placed, hsum;
    u64 off = block_excl_scan_256<u64>(nch, s_scr, &tot_ch);
    (void)block_excl_scan_256<u64>(cnt, s_scr, &placed);
    const u64 gb = block_excl_scan_256<u64>(t.ghist[d], s_scr, &hsum);
    if (t.gbase) t.gbase[d] = gb;
    if (d == 0) {
        if (t.ntiles_out) *t.ntiles_out = (u32)tot_ch;
        if (t.n_out) *t.n_out = placed;
        if (t.n == ~0ULL ? placed != hsum : (placed != t.n || hsum != t.n)) atomicOr(a.err, 8u);
    }
    const u32 *mp = t.map + (u64)d * t.vmax;
    constexpr int U = 8;                                          // map entries requested per step (a load per step is a latency per chunk)
    u64 v = 0;
    for (; v + U <= nch; v += U) {
        u32 ph[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ph[u] = mp[v + u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 left = cnt - (v + u) * CH;
            t.tile_src[off + v + u] = ((u64)(ph[u] - 1) << 32) | (left < CH ? left : CH);
        }
    }
    for (; v < nch; ++v) {
        const u64 left = cnt - v * CH;
        t.tile_src[off + v] = ((u64)(mp[v] - 1) << 32) | (left < CH ? left : CH);
    }
}

// after the second pass: every XCD must have drained its task (ticket counter past the tile count); error bit 16 otherwise
__global__ void sort_drained_kernel(const u32 *tickets, const u32 *ntiles, int n, u32 *err)
{
    const int i = threadIdx.x;
    if (i < n && tickets[i] < ntiles[i]) atomicOr(err, 16u);
}

} // namespace hsk
