// hsk_host_combine.h -- host side of the combining extraction (kernels: hsk_combine.h).
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// HSK_COMBINE=0: the instance path (expand_scatter2_kernel ...) always
static bool combine_enabled()
{
    return tune("combine", 1) != 0 && !(g_plan_flags & (HSK_FLAG_NO_AGGREGATION | HSK_FLAG_FULL_SORT | HSK_FLAG_NO_COMBINE));
}
// k-mers per bucket the bucket order aims at (a table of CB_CAP slots takes the ~400 distinct k-mers of such a bucket at 32x coverage
// with room to spare, and still most of them at 5x)
// bins of the weighted finish: the top combine_prefix_bits() key bits (8 < bits <= 16).  A bin of the instance path's 16 bits would hold
// ~120 pairs at 32x coverage and its workgroup would mostly wait for its own start-up; 14 bits: ~480 pairs on the 2048-slot table
// More pairs per task than ~900 per bin (larger genomes, lower coverage) widen the prefix for the batches that follow, up to the instance
// path's 16 bits; a bin that beats the last table before that has happened sends the call round again with 16.  HSK_COMBINE_PREFIX pins it.
constexpr int COMBINE_PREFIX_DEFAULT = 14, COMBINE_PREFIX_MAX = 16;
static int combine_prefix_forced()
{
    const long long v = tune("combine_prefix", 0);
    return v ? (int)std::min<long long>(16, std::max<long long>(9, v)) : 0;
}
static int combine_prefix_bits(const hsk_ctx *c)
{
    if (combine_prefix_forced()) return combine_prefix_forced();
    return c->combine_prefix ? c->combine_prefix : COMBINE_PREFIX_DEFAULT;
}
static int combine_prefix_for(u64 max_pairs_per_task)
{
    int p = COMBINE_PREFIX_DEFAULT;
    while (p < COMBINE_PREFIX_MAX && (max_pairs_per_task >> p) > 900) ++p;
    return p;
}
// more than one pair per combine_ratio() k-mers: the instance path (break-even measured at one per ~15); HSK_COMBINE_RATIO=1: never leave (measurements)
static u64 combine_ratio()
{
    return (u64)std::max<long long>(1, tune("combine_ratio", 16));
}
static u64 combine_bucket_kmers()
{
    return (u64)std::max<long long>(256, tune("combine_bucket", 12288));
}

static int bins_to_store(hsk_ctx *c, ScanBins &b, SupermerStore &st, u32 nvt, u32 vt_shift, hipStream_t stream)
{
    BucketItem *d_items; DALLOC(c, d_items, BucketItem *, (size_t)std::max<u32>(b.nchunks, 1) * sizeof(BucketItem));
    if (b.nchunks) {
        BinItemsArgs ba; ba.cursor = b.cursor; ba.map = b.map; ba.vmax = b.vmax; ba.chunk_bin = b.chunk_bin; ba.nchunks = b.nchunks; ba.nvt = nvt; ba.vt_shift = vt_shift; ba.items = d_items;
        hipLaunchKernelGGL(bins_items_kernel, dim3((b.nchunks + 255) / 256), dim3(256), 0, stream, ba);
        HIPCHK(c, hipGetLastError());
    }
    st.sm_item = reinterpret_cast<u64 *>(b.items); st.sm_sub = b.subs; st.d_bitems = d_items; st.n_bitems = b.nchunks;
    st.bin_aux[0] = b.cursor; st.bin_aux[1] = b.map; st.bin_aux[2] = b.ctl; st.bin_aux[3] = b.chunk_bin; c->pool.release(b.d_table);
    b = ScanBins();                                         // (the store owns everything now)
    return HSK_OK;
}

// the supermer items of the owned tasks in bucket order
struct BucketOrder {
    ulonglong2 *recs = nullptr; u32 *off = nullptr, *cur = nullptr, *d_log2nb = nullptr; u64 *d_out_base = nullptr; BucketItem *d_items = nullptr;
    uint2 *units = nullptr; u64 *d_unit_off = nullptr; u32 *d_nunits = nullptr;      // the work units of the combining extraction (bucket_units_kernel)
    u32 stride = 0;
    std::vector<u32> log2nb; std::vector<u64> out_base, unit_off;
    bool active = false;
};
static void bucket_release(hsk_ctx *c, BucketOrder &bo)
{
    c->pool.release(bo.recs); c->pool.release(bo.off); c->pool.release(bo.cur); c->pool.release(bo.d_log2nb); c->pool.release(bo.d_out_base); c->pool.release(bo.d_items);
    c->pool.release(bo.units); c->pool.release(bo.d_unit_off); c->pool.release(bo.d_nunits);
    bo = BucketOrder();
}

// tasks: the owned tasks (EMPTY entries ~0u skipped).  Returns HSK_OK with bo.active = false when a task does not fit 32-bit record offsets.
static int bucket_order_tasks(hsk_ctx *c, u32 ntasks, const std::vector<TaskSegs> &segs, const std::vector<u32> &tasks, const BaseSource &src, u32 vt_shift, BucketOrder &bo)
{
    bo = BucketOrder();
    bo.log2nb.assign(ntasks, 0); bo.out_base.assign(ntasks, 0); bo.unit_off.assign(ntasks, 0);
    std::vector<BucketItem> items;
    u64 run = 0, urun = 0; u32 maxlg = 0;
    const u64 target = combine_bucket_kmers();
    for (u32 t : tasks) {
        if (t == ~0u) continue;
        u64 nsup = 0;
        for (const ExpSeg &sg : segs[t].segs) {
            for (u64 o = 0; o < sg.n_sup; o += CS_ITEM) { BucketItem it; it.first = sg.sup_off + o; it.n = (u32)std::min<u64>(CS_ITEM, sg.n_sup - o); it.task = (u16)t; it.hi = vt_shift ? (u16)sg.byte_off : (u16)0; items.push_back(it); }      // (without virtual tasks byte_off is a real byte offset)
            nsup += sg.n_sup;
        }
        if (nsup >= (1ULL << 32)) return HSK_OK;
        u32 lg = 0;
        while (lg < (u32)CS_MAX_LOG2NB && lg < (u32)CS_MAX_LOCAL + vt_shift && (segs[t].nkmers >> lg) > target) ++lg;
        bo.log2nb[t] = lg; maxlg = std::max(maxlg, lg);
        bo.out_base[t] = run; run += nsup;
        bo.unit_off[t] = urun; urun += (1ULL << lg) + nsup / CB_UNIT + 1;      // (a bucket makes at most one unit more than it has whole slices)
    }
    const bool dev_list = src.bitems != nullptr;             // scan-placed bins: the work list is on the device already (one item per chunk)
    if (dev_list ? src.n_bitems == 0 : items.empty()) return HSK_OK;
    const u32 nwork = dev_list ? src.n_bitems : (u32)items.size();
    bo.stride = (1u << maxlg) + 1;
    DALLOC(c, bo.recs, ulonglong2 *, run * 16 + 64);
    DALLOC(c, bo.off, u32 *, (size_t)ntasks * bo.stride * 4);
    DALLOC(c, bo.cur, u32 *, (size_t)ntasks * bo.stride * 4);
    DALLOC(c, bo.d_log2nb, u32 *, (size_t)ntasks * 4);
    DALLOC(c, bo.d_out_base, u64 *, (size_t)ntasks * 8);
    if (!dev_list) DALLOC(c, bo.d_items, BucketItem *, items.size() * sizeof(BucketItem));
    DALLOC(c, bo.units, uint2 *, urun * 8 + 64); DALLOC(c, bo.d_unit_off, u64 *, (size_t)ntasks * 8); DALLOC(c, bo.d_nunits, u32 *, (size_t)ntasks * 4);
    HIPCHK(c, hipMemsetAsync(bo.d_nunits, 0, (size_t)ntasks * 4, c->stream));
    HIPCHK(c, hipMemcpyAsync(bo.d_unit_off, bo.unit_off.data(), (size_t)ntasks * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(bo.off, 0, (size_t)ntasks * bo.stride * 4, c->stream));
    HIPCHK(c, hipMemcpyAsync(bo.d_log2nb, bo.log2nb.data(), (size_t)ntasks * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(bo.d_out_base, bo.out_base.data(), (size_t)ntasks * 8, hipMemcpyHostToDevice, c->stream));
    if (!dev_list) HIPCHK(c, hipMemcpyAsync(bo.d_items, items.data(), items.size() * sizeof(BucketItem), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));                  // (the item list and the small tables are host memory of this function)
    BucketSortArgs a; memset(&a, 0, sizeof a);
    a.items = dev_list ? reinterpret_cast<const BucketItem *>(src.bitems) : bo.d_items; a.sm_sub = src.sub; a.sm_item = reinterpret_cast<const ulonglong2 *>(src.item); a.off = bo.off; a.cur = bo.cur; a.log2nb = bo.d_log2nb; a.out_base = bo.d_out_base;
    a.stride = bo.stride; a.recs = bo.recs; a.vt_shift = vt_shift; a.err = c->d_err;
    a.units = bo.units; a.unit_off = bo.d_unit_off; a.nunits = bo.d_nunits;
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 8; ep.keys = run; (void)hipEventRecord(ep.a, c->stream); }
    hipLaunchKernelGGL(bucket_hist_kernel, dim3(nwork), dim3(CS_THREADS), 0, c->stream, a);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(ntasks), dim3(1024), 0, c->stream, a);
    hipLaunchKernelGGL(bucket_units_kernel, dim3(ntasks), dim3(1024), 0, c->stream, a);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(nwork), dim3(CS_THREADS), 0, c->stream, a);
    if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
    HIPCHK(c, hipGetLastError());
    bo.active = true;
    return HSK_OK;
}

// Several ranks: the items of a batch of owned tasks, built from the supermers the exchange delivered (jobs[i]: segments, len[], byte runs;
// sub16[i]: their minimizer bits), grouped by (task, virtual task).  gsegs[t] / gsrc then describe an item-mode store of these tasks for
// bucket_order_tasks (16 virtual tasks per task: vt_shift 4).  One wait (the runs' sizes come back to the host).
struct FedItems { ulonglong2 *items = nullptr; u32 *subs = nullptr, *vt_cnt = nullptr; u64 *vt_cur = nullptr; ExpandScratch x[XCD_BATCH]; };
constexpr u32 FED_VT_SHIFT = 4;
static void fed_release(hsk_ctx *c, FedItems &f)
{
    c->pool.release(f.items); c->pool.release(f.subs); c->pool.release(f.vt_cnt); c->pool.release(f.vt_cur);
    for (int i = 0; i < XCD_BATCH; ++i) expand_release(c, f.x[i]);
    f = FedItems();
}
static int build_items_batch(hsk_ctx *c, u32 ntasks, const u32 *tk, const ExpandJob *jobs, const unsigned short *const *sub16, std::vector<TaskSegs> &gsegs,
                             BaseSource &gsrc, FedItems &f, hipStream_t stream)
{
    f = FedItems();
    gsegs.assign(ntasks, TaskSegs());
    const TaskSegs *tsp[XCD_BATCH]; const u8 *lens[XCD_BATCH]; int idx[XCD_BATCH], m = 0;
    u64 total = 0, max_tiles = 0;
    for (int i = 0; i < XCD_BATCH; ++i) {
        if (tk[i] == ~0u || !jobs[i].ts->ntiles) continue;
        if (!sub16[i]) return fail(c, HSK_ERR_INTERNAL, "a received task without minimizer bits");
        tsp[m] = jobs[i].ts; lens[m] = jobs[i].sm_len; idx[m] = i; ++m;
        for (const ExpSeg &sg : jobs[i].ts->segs) total += sg.n_sup;
        max_tiles = std::max(max_tiles, jobs[i].ts->ntiles);
    }
    if (!m) return HSK_OK;
    if (total >= (1ULL << 32)) return fail(c, HSK_ERR_UNSUPPORTED, "a batch of more than 2^32 supermers");
    int rc = expand_prepare_batch(c, m, tsp, lens, f.x, stream, false, true); if (rc) return rc;
    DALLOC(c, f.items, ulonglong2 *, total * 16 + 64); DALLOC(c, f.subs, u32 *, total * 4 + 64);
    DALLOC(c, f.vt_cnt, u32 *, 128 * 4); DALLOC(c, f.vt_cur, u64 *, 128 * 8);
    HIPCHK(c, hipMemsetAsync(f.vt_cnt, 0, 128 * 4, stream));
    ItemBuildArgs a; memset(&a, 0, sizeof a);
    for (int j = 0; j < m; ++j) {
        const ExpandJob &jb = jobs[idx[j]];
        ItemBuildTask &t = a.t[j];
        t.segs = f.x[j].d_segs; t.nseg = (int)jb.ts->segs.size(); t.sm_len = jb.sm_len; t.sub16 = sub16[idx[j]]; t.src8 = jb.src.src8; t.src_words = jb.src.nwords;
        t.tile_off = f.x[j].d_tile_off; t.ntiles = jb.ts->ntiles;
    }
    a.vt_cnt = f.vt_cnt; a.vt_cur = f.vt_cur; a.items = f.items; a.subs = f.subs; a.k = c->cfg.kmer_size; a.err = c->d_err;
    hipLaunchKernelGGL(vt_hist_kernel, dim3((u32)std::min<u64>((max_tiles + 3) / 4, 1024), m), dim3(EXP_THREADS), 0, stream, a);
    hipLaunchKernelGGL(vt_scan_kernel, dim3(1), dim3(128), 0, stream, a);
    u32 *h_cnt = (u32 *)((char *)c->pinned + (320u << 10));
    HIPCHK(c, hipMemcpyAsync(h_cnt, f.vt_cnt, 128 * 4, hipMemcpyDeviceToHost, stream));
    hipLaunchKernelGGL(items_build_kernel, dim3((u32)((max_tiles + IB_TILES - 1) / IB_TILES), m), dim3(EXP_THREADS), 0, stream, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hsk_sync(c, stream));
    u64 run = 0;
    for (int j = 0; j < m; ++j) {
        const u32 t = tk[idx[j]];
        TaskSegs &ts = gsegs[t];
        ts.nkmers = jobs[idx[j]].ts->nkmers;
        for (u32 v = 0; v < 16; ++v) {
            const u32 n = h_cnt[j * 16 + v];
            if (n) { ExpSeg sg; sg.sup_off = run; sg.n_sup = n; sg.byte_off = v; sg.kmer_off = 0; sg.tile_start = 0; ts.segs.push_back(sg); }
            run += n;
        }
    }
    if (run != total) return fail(c, HSK_ERR_INTERNAL, "item build: %llu of %llu supermers placed", (unsigned long long)run, (unsigned long long)total);
    gsrc = BaseSource(); gsrc.sub = f.subs; gsrc.item = reinterpret_cast<const u64 *>(f.items);
    return HSK_OK;
}

// The batch's tasks (tk[i]: task of XCD i, ~0u: none) from bucket-ordered records to {k-mer, count} pairs in the chunk stores bt[i].kB
// (keys) / bt[i].vB (counts); the histogram of the second pass's digit goes to ghist[i] + 256.  What follows is sort_batch_prescattered
// with the counts as payload; sb.h_nout[i] then holds the pairs of task i (read after the stream has passed chunk_tiles_kernel).
// pair_cap: records the chunk stores bt[i].kB / vB hold (scatter_store_keys(pair_cap) + one chunk that takes what does not fit)
template <int NW>
static int combine_batch(hsk_ctx *c, const u32 *tk, const BatchTask *bt, u64 *const *ghist, const PassDesc *plan, const BucketOrder &bo,
                         u64 *h_nout, ScatterBatch &sb, hipStream_t stream, u64 pair_cap)
{
    static_assert(NW == 1 || NW == 2, "keys of one or two words");
    constexpr int CH = XsCfg<NW>::CHUNK;
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    sb = ScatterBatch();
    ScatterArgs &a = sb.args; memset(&a, 0, sizeof a);
    CombineArgs ca; memset(&ca, 0, sizeof ca);
    bool any = false;
    for (int i = 0; i < XCD_BATCH; ++i) if (tk[i] != ~0u && bt[i].n) any = true;
    if (!any) return HSK_OK;
    DALLOC(c, sb.d_cursor, u64 *, (size_t)XCD_BATCH * 256 * 8);
    DALLOC(c, sb.d_ctl, u32 *, (size_t)XCD_BATCH * 16);
    DALLOC(c, sb.d_gbase, u64 *, (size_t)XCD_BATCH * 256 * 8);
    DALLOC(c, sb.d_ntiles, u32 *, 256);
    DALLOC(c, sb.d_nout, u64 *, 256);
    sb.h_nout = h_nout;
    HIPCHK(c, hipMemsetAsync(sb.d_ntiles, 0, 64, stream));
    HIPCHK(c, hipMemsetAsync(sb.d_nout, 0, 64, stream));
    HIPCHK(c, hipMemsetAsync(sb.d_cursor, 0, (size_t)XCD_BATCH * 256 * 8, stream));
    HIPCHK(c, hipMemsetAsync(sb.d_ctl, 0, (size_t)XCD_BATCH * 16, stream));
    u64 ntot = 0;
    for (int i = 0; i < XCD_BATCH; ++i) {
        if (tk[i] == ~0u || !bt[i].n) continue;
        const u32 tid = tk[i];
        const u64 n = bt[i].n;                           // k-mers of the task: never fewer than its pairs
        ScatterTask &t = a.t[i];
        t.vmax = (u32)(n / CH + 1);
        DALLOC(c, sb.d_map[i], u32 *, (size_t)256 * t.vmax * 4);
        DALLOC(c, sb.d_tile_src[i], u64 *, (size_t)(n / CH + 257) * 8);
        HIPCHK(c, hipMemsetAsync(sb.d_map[i], 0, (size_t)256 * t.vmax * 4, stream));
        t.ntiles = 1;                                    // (chunk_tiles_kernel: the XCD has a task)
        t.chunks = bt[i].kB; t.vchunks = bt[i].vB; t.cursor = sb.d_cursor + (size_t)i * 256; t.map = sb.d_map[i]; t.ctl = sb.d_ctl + (size_t)i * 4;
        t.ghist = ghist[i] + 256; t.tile_src = sb.d_tile_src[i];
        t.n = ~0ULL; t.n_out = sb.d_nout + i; t.gbase = sb.d_gbase + (size_t)i * 256; t.ntiles_out = sb.d_ntiles + i;
        CombineTask &q = ca.t[i];
        q.recs = bo.recs + bo.out_base[tid]; q.units = bo.units + bo.unit_off[tid]; q.nunits = bo.d_nunits + tid; q.nb = 1u << bo.log2nb[tid]; q.vmax = t.vmax;
        q.cap_chunks = (u32)(scatter_store_keys(pair_cap, CH) / CH);
        q.chunks = t.chunks; q.vchunks = t.vchunks; q.cursor = t.cursor; q.map = t.map; q.ctl = t.ctl; q.ghist = t.ghist;
        ntot += n;
    }
    a.k = c->cfg.kmer_size; a.shift0 = plan[0].shift; a.shift1 = plan[1].shift; a.chunk = CH; a.err = c->d_err;
    ca.k = a.k; ca.shift0 = a.shift0; ca.bits0 = plan[0].bits; ca.shift1 = a.shift1; ca.err = c->d_err;
    static int occ = 0;
    if (!occ) {
        int nb = 0;
        const hipError_t e = NW == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, combine_kernel<31>, CB_THREADS, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, combine2_kernel<51>, CB_THREADS, 0);
        occ = (e == hipSuccess && nb > 0) ? nb : 2;
    }
    EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 7; ep.keys = ntot; ep.bytes = 0; (void)hipEventRecord(ep.a, stream); }
    const u32 grid = (u32)occ * 256u;
    if (NW == 2) { if (a.k == 51) hipLaunchKernelGGL((combine2_kernel<51>), dim3(grid), dim3(CB_THREADS), 0, stream, ca); else hipLaunchKernelGGL((combine2_kernel<0>), dim3(grid), dim3(CB_THREADS), 0, stream, ca); }
    else if (a.k == 31) hipLaunchKernelGGL((combine_kernel<31>), dim3(grid), dim3(CB_THREADS), 0, stream, ca);
    else hipLaunchKernelGGL((combine_kernel<0>), dim3(grid), dim3(CB_THREADS), 0, stream, ca);
    if (profile) { (void)hipEventRecord(ep.b, stream); c->ev_pending.push_back(ep); }
    // the tile lists of the second pass and the pairs of every task (the host sizes the second pass and the finish from them)
    hipLaunchKernelGGL(chunk_tiles_kernel, dim3(XCD_BATCH), dim3(256), 0, stream, sb.args);
    HIPCHK(c, hipMemcpyAsync(sb.h_nout, sb.d_nout, XCD_BATCH * 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(c, hipMemcpyAsync(sb.h_nout + XCD_BATCH, c->d_err, 4, hipMemcpyDeviceToHost, stream));      // (bit 512: the pair stores ran over)
    HIPCHK(c, hipGetLastError());
    sb.active = true; sb.tiles_done = true;
    return HSK_OK;
}
