// hsk_host_scatter.h -- host side of the expand fused with the first scatter pass (kernels: hsk_scatter.h).
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// device state of one batch between expand_scatter_kernel and the second pass
struct ScatterBatch {
    ScatterArgs args;
    u64 *d_cursor = nullptr;                     // [XCD_BATCH][256]
    u32 *d_ctl = nullptr;                        // [XCD_BATCH][4]
    u32 *d_map[XCD_BATCH] = {nullptr};
    u64 *d_tile_src[XCD_BATCH] = {nullptr};
    u64 *d_gbase = nullptr;                      // [XCD_BATCH][256] digit bases of the second pass (chunk_tiles_kernel)
    u32 *d_ntiles = nullptr;                     // [XCD_BATCH] tiles of the second pass (chunk_tiles_kernel)
    u64 *d_nout = nullptr; u64 *h_nout = nullptr; // combining extraction: [XCD_BATCH] pairs in every task's chunk store (device word / pinned copy asked for behind chunk_tiles_kernel)
    ExpandScratch x[EXP_BATCH];                  // segment lists (and tile offsets) the kernel reads
    bool active = false;
    bool tiles_done = false;                     // chunk_tiles_kernel has run already (combining extraction: the host reads the pair counts before the second pass is sized)
};

static bool scatter_enabled()
{
    return tune("fused_scatter", 1) != 0;
}
// keys the chunk store of a task of n k-mers must hold (less than one chunk is wasted per digit)
static size_t scatter_store_keys(u64 n, int chunk) { return (size_t)(n / chunk + 257) * chunk; }

static void scatter_release(hsk_ctx *c, ScatterBatch &sb)
{
    c->pool.release(sb.d_cursor); c->pool.release(sb.d_ctl); c->pool.release(sb.d_gbase); c->pool.release(sb.d_ntiles); c->pool.release(sb.d_nout);
    for (int i = 0; i < XCD_BATCH; ++i) { c->pool.release(sb.d_map[i]); c->pool.release(sb.d_tile_src[i]); expand_release(c, sb.x[i]); }
    sb = ScatterBatch();
}

// jobs[i] is the task XCD i expands (ts->ntiles == 0: none); its keys go to the chunk store jobs[i].keys, the histogram
// of the second pass's digit to jobs[i].ghist + 256.  plan: the two 8-bit passes of the prefix plan.
template <int NW>
static int scatter_expand_batch(hsk_ctx *c, const ExpandJob *jobs, const BatchTask *bt, const PassDesc *plan, ScatterBatch &sb, hipStream_t stream)
{
    constexpr int XS_CHUNK = XsCfg<NW>::CHUNK;
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    sb = ScatterBatch();
    ScatterArgs &a = sb.args; memset(&a, 0, sizeof a);
    ExpandScratch *x = sb.x; int xi[EXP_BATCH]; int m = 0;
    const TaskSegs *tsp[EXP_BATCH]; const u8 *lens[EXP_BATCH];
    bool offsets = false;
    for (int i = 0; i < XCD_BATCH; ++i) {
        xi[i] = -1;
        if (!jobs[i].ts->ntiles) continue;
        tsp[m] = jobs[i].ts; lens[m] = jobs[i].sm_len; xi[i] = m++;
        if (!reads_in_place(jobs[i].src)) offsets = true;           // byte streams: a tile's input offset is a prefix over the tiles before it
    }
    if (m == 0) return HSK_OK;
    int rc = expand_prepare_batch(c, m, tsp, lens, x, stream, false, offsets); if (rc) return rc;
    DALLOC(c, sb.d_cursor, u64 *, (size_t)XCD_BATCH * 256 * 8);
    DALLOC(c, sb.d_ctl, u32 *, (size_t)XCD_BATCH * 16);
    DALLOC(c, sb.d_gbase, u64 *, (size_t)XCD_BATCH * 256 * 8);
    DALLOC(c, sb.d_ntiles, u32 *, 256);
    HIPCHK(c, hipMemsetAsync(sb.d_ntiles, 0, 64, stream));
    HIPCHK(c, hipMemsetAsync(sb.d_cursor, 0, (size_t)XCD_BATCH * 256 * 8, stream));
    HIPCHK(c, hipMemsetAsync(sb.d_ctl, 0, (size_t)XCD_BATCH * 16, stream));
    u64 ntot = 0;
    for (int i = 0; i < XCD_BATCH; ++i) {
        if (xi[i] < 0) continue;
        const ExpandJob &j = jobs[i];
        ScatterTask &t = a.t[i];
        const u64 n = bt[i].n;
        t.vmax = (u32)(n / XS_CHUNK + 1);
        DALLOC(c, sb.d_map[i], u32 *, (size_t)256 * t.vmax * 4);
        DALLOC(c, sb.d_tile_src[i], u64 *, (size_t)(n / XS_CHUNK + 257) * 8);
        HIPCHK(c, hipMemsetAsync(sb.d_map[i], 0, (size_t)256 * t.vmax * 4, stream));
        t.segs = x[xi[i]].d_segs; t.nseg = (int)j.ts->segs.size(); t.sm_len = j.sm_len;
        t.src8 = j.src.src8; t.src_bit0 = j.src.bit0; t.src_words = j.src.nwords; t.sm_gpos = j.src.gpos; t.sm_boff = j.src.boff;
        t.tile_off = offsets ? x[xi[i]].d_tile_off : nullptr; t.ntiles = j.ts->ntiles;
        t.chunks = j.keys; t.cursor = sb.d_cursor + (size_t)i * 256; t.map = sb.d_map[i]; t.ctl = sb.d_ctl + (size_t)i * 4;
        t.ghist = j.ghist + 256; t.tile_src = sb.d_tile_src[i];
        t.n = n; t.gbase = sb.d_gbase + (size_t)i * 256; t.ntiles_out = sb.d_ntiles + i;
        t.sm_pos = j.sm_pos; t.sm_rid = j.sm_rid; t.vchunks = j.vals;
        ntot += n;
    }
    a.k = c->cfg.kmer_size; a.shift0 = plan[0].shift; a.shift1 = plan[1].shift; a.chunk = XS_CHUNK; a.err = c->d_err;
    const bool ext = c->cfg.extension != 0;
    // expand_scatter2_kernel (two sweeps per flush, three workgroups per CU: 33.0 against 36.2 ms per step on the benchmark) takes
    // one-word keys without payload whose bases are read in place, at most XS_MAXSEG segments per task; HSK_XS2=0: the one-sweep kernel
    const bool xs2_env = tune("xs2", 1) != 0;
    bool xs2 = xs2_env && NW == 1 && !ext;
    for (int i = 0; i < XCD_BATCH && xs2; ++i) if (xi[i] >= 0 && !(reads_in_place(jobs[i].src) && jobs[i].ts->segs.size() <= (size_t)XS_MAXSEG)) xs2 = false;
    if constexpr (NW == 1) {
        if (xs2) {
            static int occ2 = 0;
            if (!occ2) { int nb = 0; occ2 = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, expand_scatter2_kernel<31>, XS_THREADS, 0) == hipSuccess && nb > 0) ? nb : 2; }
            EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 1; ep.keys = ntot; ep.bytes = ntot * 8; (void)hipEventRecord(ep.a, stream); }
            const u32 grid = (u32)occ2 * 256u;
            if (a.k == 31 && a.shift0 == 48 && a.shift1 == 56) hipLaunchKernelGGL((expand_scatter2_kernel<31>), dim3(grid), dim3(XS_THREADS), 0, stream, a);
            else hipLaunchKernelGGL((expand_scatter2_kernel<0>), dim3(grid), dim3(XS_THREADS), 0, stream, a);
            if (profile) { (void)hipEventRecord(ep.b, stream); c->ev_pending.push_back(ep); }
            HIPCHK(c, hipGetLastError());
            sb.active = true;
            return HSK_OK;
        }
    }
    static int occ_c[2] = {0, 0};                             // (per instantiation of this template: per NW)
    int &occ = occ_c[ext ? 1 : 0];
    if (!occ) {
        int nb = 0;
        hipError_t e;
        if constexpr (NW == 1) e = ext ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, expand_scatter_kernel<1, true>, XS_THREADS, 0)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, expand_scatter_kernel<1, false>, XS_THREADS, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, expand_scatter_kernel<NW, false>, XS_THREADS, 0);
        occ = (e == hipSuccess && nb > 0) ? nb : 2;
    }
    EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 1; ep.keys = ntot; ep.bytes = ntot * (ext ? 16 : 8 * NW); (void)hipEventRecord(ep.a, stream); }
    const u32 grid = (u32)occ * 256u;
    const bool xs_generic = tune("scatter_generic", 0) != 0;        // (tests: the default k through the generic instance)
    if constexpr (NW == 1) {
        if (ext) hipLaunchKernelGGL((expand_scatter_kernel<1, true>), dim3(grid), dim3(XS_THREADS), 0, stream, a);
        else if (a.k == 31 && a.shift0 == 48 && a.shift1 == 56 && !xs_generic) hipLaunchKernelGGL((expand_scatter_kernel<1, false, 31>), dim3(grid), dim3(XS_THREADS), 0, stream, a);
        else hipLaunchKernelGGL((expand_scatter_kernel<1, false>), dim3(grid), dim3(XS_THREADS), 0, stream, a);
    } else if (NW == 2 && a.k == 51 && a.shift0 == 48 && a.shift1 == 56 && !xs_generic) hipLaunchKernelGGL((expand_scatter_kernel<NW, false, NW == 2 ? 51 : 0>), dim3(grid), dim3(XS_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((expand_scatter_kernel<NW, false>), dim3(grid), dim3(XS_THREADS), 0, stream, a);
    if (profile) { (void)hipEventRecord(ep.b, stream); c->ev_pending.push_back(ep); }
    HIPCHK(c, hipGetLastError());
    sb.active = true;
    return HSK_OK;
}

// Second (stable) pass over a batch whose first pass was done by expand_scatter_kernel: input = the chunk stores bt[i].kB,
// output = bt[i].kA.  Nothing is read back by the host: chunk_tiles_kernel derives the digit bases, the tile list and the
// tile count on the device (and checks that every XCD expanded its task), the look-back table and the grid are sized for
// the most tiles a task of n keys can have (n / CHUNK + 256: every digit wastes less than one chunk), workgroups beyond the
// real tile count leave at once, and sort_drained_kernel checks afterwards that every XCD drained its task.  A failed
// check sets the sticky device error word, which the caller reads once per hsk_count (check_device_error).
template <int NW>
static int sort_batch_prescattered(hsk_ctx *c, BatchTask *bt, const PassDesc *plan, u64 * /*d_ghist*/, ScatterBatch &sb)
{
    constexpr int XS_CHUNK = XsCfg<NW>::CHUNK;
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    const bool has_val = bt[0].vA != nullptr;
    for (int i = 0; i < XCD_BATCH; ++i) { bt[i].out_k = bt[i].kA; bt[i].out_v = has_val ? bt[i].vA : nullptr; }
    if (!sb.active) return HSK_OK;
    u64 tile_cap[XCD_BATCH], max_tiles = 0, ntot = 0; bool wide = force_wide_lookback();
    size_t lb_off[XCD_BATCH + 1]; lb_off[0] = 0;
    for (int i = 0; i < XCD_BATCH; ++i) {
        tile_cap[i] = bt[i].n ? bt[i].n / XS_CHUNK + 256 : 0;
        if (bt[i].n >= (1ULL << 30)) wide = true;
        max_tiles = std::max(max_tiles, tile_cap[i]); ntot += bt[i].n;
    }
    const size_t lbw = wide ? 8 : 4;
    for (int i = 0; i < XCD_BATCH; ++i) lb_off[i + 1] = lb_off[i] + (size_t)tile_cap[i] * 256 * lbw;
    u32 *d_tickets; void *d_lookback;
    DALLOC(c, d_tickets, u32 *, 256);
    DALLOC(c, d_lookback, void *, lb_off[XCD_BATCH] + 256);
    HIPCHK(c, hipMemsetAsync(d_tickets, 0, 64, c->stream));
    HIPCHK(c, hipMemsetAsync(d_lookback, 0, lb_off[XCD_BATCH], c->stream));
    if (!sb.tiles_done) hipLaunchKernelGGL(chunk_tiles_kernel, dim3(XCD_BATCH), dim3(256), 0, c->stream, sb.args);
    MultiSortArgs ms; memset(&ms, 0, sizeof ms);
    for (int i = 0; i < XCD_BATCH; ++i) {
        SortArgs &a = ms.t[i];
        a.keys_in = bt[i].kB; a.keys_out = bt[i].kA; a.vals_in = has_val ? bt[i].vB : nullptr; a.vals_out = has_val ? bt[i].vA : nullptr; a.n = bt[i].n;
        a.ntiles = tile_cap[i]; a.ntiles_dev = sb.d_ntiles + i;
        a.word = plan[1].word; a.shift = plan[1].shift; a.bits = plan[1].bits;
        a.unstable = unstable_first_pass() ? 1 : 0;     // a tile is a chunk of ONE first-pass digit: the order inside it is free, the look-back keeps the tiles in order
        a.gbase = sb.d_gbase + (size_t)i * 256; a.lookback = (char *)d_lookback + lb_off[i];
        a.ticket = d_tickets + i; a.err = c->d_err; a.tile_src = sb.d_tile_src[i];
    }
    // the real tile count is n / CHUNK + (digits with a partial chunk) <= the cap; the grid follows the cap (+12 %: the
    // dispatcher deals workgroups round-robin over the XCDs, not exactly evenly)
    const u32 grid = (u32)(XCD_BATCH * (max_tiles + max_tiles / 8) + 64);
    EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 0; ep.keys = ntot; ep.bytes = 2 * ntot * (has_val ? 16 : 8 * NW); (void)hipEventRecord(ep.a, c->stream); }
    if (max_tiles) {
        if (has_val) { if (wide) launch_onesweep_multi<NW, true, u64>(c, ms, grid); else launch_onesweep_multi<NW, true, u32>(c, ms, grid); }
        else { if (wide) launch_onesweep_multi<NW, false, u64>(c, ms, grid); else launch_onesweep_multi<NW, false, u32>(c, ms, grid); }
    }
    if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
    hipLaunchKernelGGL(sort_drained_kernel, dim3(1), dim3(64), 0, c->stream, d_tickets, sb.d_ntiles, XCD_BATCH, c->d_err);
    HIPCHK(c, hipGetLastError());
    // released without a wait: every later user of these blocks is enqueued on the same stream
    c->pool.release(d_tickets); c->pool.release(d_lookback);
    scatter_release(c, sb);
    return HSK_OK;
}
