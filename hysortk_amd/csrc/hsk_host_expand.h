// hsk_host_expand.h -- host side of the expand stage and of the byte packing for the exchange (kernels: hsk_expand.h).
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// ------------------------------------------------------------------------------------------------
// stage: expand one task (a11)
// ------------------------------------------------------------------------------------------------
static void finalize_segs(TaskSegs &ts)
{
    u64 tile = 0;
    for (auto &s : ts.segs) { s.tile_start = tile; tile += (s.n_sup + EXP_TILE - 1) / EXP_TILE; }
    ts.ntiles = tile;
}

struct ExpandScratch { ExpSeg *d_segs = nullptr; u64 *d_tile_sum = nullptr, *d_tile_off = nullptr, *d_cursor = nullptr; };

// tile sums + scans of n <= EXP_PREP_BATCH tasks with two launches
// offsets = false: only the segment lists are uploaded (tiles then reserve their output ranges themselves)
static int expand_prepare_batch(hsk_ctx *c, int n, const TaskSegs *const *ts, const u8 *const *sm_len, ExpandScratch *x,
                                hipStream_t stream = nullptr, bool prealloc = false, bool offsets = true)
{
    if (!stream) stream = c->stream;
    ExpandPrepArgs pa; memset(&pa, 0, sizeof pa);
    pa.k = c->cfg.kmer_size;
    u64 max_tiles = 0; int max_seg = 0;
    for (int i = 0; i < n; ++i) {
        const int nseg = (int)ts[i]->segs.size();
        if (!prealloc) {
            DALLOC(c, x[i].d_segs, ExpSeg *, sizeof(ExpSeg) * nseg);
            if (offsets) { DALLOC(c, x[i].d_tile_sum, u64 *, ts[i]->ntiles * 16); DALLOC(c, x[i].d_tile_off, u64 *, ts[i]->ntiles * 16); }
        }
        HIPCHK(c, hipMemcpyAsync(x[i].d_segs, ts[i]->segs.data(), sizeof(ExpSeg) * nseg, hipMemcpyHostToDevice, stream));
        pa.segs[i] = x[i].d_segs; pa.nseg[i] = nseg; pa.sm_len[i] = sm_len[i]; pa.ntiles[i] = ts[i]->ntiles;
        pa.tile_sum[i] = x[i].d_tile_sum; pa.tile_off[i] = x[i].d_tile_off;
        max_tiles = std::max(max_tiles, ts[i]->ntiles); max_seg = std::max(max_seg, nseg);
    }
    if (max_tiles == 0 || !offsets) return HSK_OK;
    hipLaunchKernelGGL(expand_tilesum_kernel, dim3((u32)((max_tiles + EXP_SUM_TILES - 1) / EXP_SUM_TILES), n), dim3(EXP_THREADS), 0, stream, pa);
    hipLaunchKernelGGL(expand_scan_kernel, dim3(max_seg, n), dim3(EXP_THREADS), 0, stream, pa);
    return HSK_OK;
}
static int expand_prepare(hsk_ctx *c, const TaskSegs &ts, const u8 *sm_len, ExpandScratch &x)
{
    const TaskSegs *tp = &ts;
    return expand_prepare_batch(c, 1, &tp, &sm_len, &x);
}
static void expand_release(hsk_ctx *c, ExpandScratch &x) { c->pool.release(x.d_segs); c->pool.release(x.d_tile_sum); c->pool.release(x.d_tile_off); c->pool.release(x.d_cursor); x = ExpandScratch(); }

// One launch for up to EXP_BATCH tasks (hsk_expand.h).  ghist[i] (optional) receives the digit histograms of
// the `npass` radix passes in `plan` for task i.
struct ExpandJob { const TaskSegs *ts; const u8 *sm_len; BaseSource src; const u32 *sm_pos; const int32_t *sm_rid; u64 *keys, *vals, *ghist; };

template <int NW>
static int expand_batch(hsk_ctx *c, const ExpandJob *jobs, int njobs, int npass = 0, const PassDesc *plan = nullptr,
                        hipStream_t stream = nullptr, ExpandScratch *pre = nullptr)
{
    const bool ext = c->cfg.extension != 0;
    if (!stream) stream = c->stream;
    ExpandArgs a; memset(&a, 0, sizeof a);
    ExpandScratch xown[EXP_BATCH];
    ExpandScratch *x = pre ? pre : xown;
    int nt = 0; u64 max_tiles = 0;
    const bool reserve_enabled = tune("expand_reserve", 1) != 0;
    bool reserve = true;
    {
        const TaskSegs *tsp[EXP_BATCH]; const u8 *lens[EXP_BATCH]; int m = 0;
        for (int i = 0; i < njobs; ++i) if (jobs[i].ts->ntiles) { tsp[m] = jobs[i].ts; lens[m] = jobs[i].sm_len; ++m; }
        // (ts.segs is host memory owned by the caller and stays alive until the next sync)
        // bases read in place (one GPU, one segment per task): no tile sums, a tile takes its output range with an atomic
        for (int i = 0; i < njobs; ++i) if (jobs[i].ts->ntiles && (!reads_in_place(jobs[i].src) || jobs[i].ts->segs.size() != 1)) reserve = false;
        if (!reserve_enabled) reserve = false;
        int rc = expand_prepare_batch(c, m, tsp, lens, x, stream, pre != nullptr, !reserve); if (rc) return rc;
    }
    u64 *d_kcur = nullptr;
    if (reserve) {
        d_kcur = pre ? pre[0].d_cursor : (u64 *)c->pool.alloc(256);
        if (!d_kcur) return fail(c, HSK_ERR_OOM, "expand cursors");
        HIPCHK(c, hipMemsetAsync(d_kcur, 0, EXP_BATCH * 8, stream));
    }
    for (int i = 0; i < njobs; ++i) {
        const ExpandJob &j = jobs[i];
        if (j.ts->ntiles == 0) continue;
        ExpandTask &t = a.t[nt];
        t.segs = x[nt].d_segs; t.nseg = (int)j.ts->segs.size(); t.sm_len = j.sm_len;
        t.src8 = j.src.src8; t.src_bit0 = j.src.bit0; t.src_words = j.src.nwords; t.sm_gpos = j.src.gpos; t.sm_boff = j.src.boff; t.sm_pos = j.sm_pos; t.sm_rid = j.sm_rid;
        t.tile_off = reserve ? nullptr : x[nt].d_tile_off; t.kcursor = reserve ? d_kcur + nt : nullptr; t.ntiles = j.ts->ntiles; t.keys_out = j.keys; t.vals_out = j.vals; t.ghist = npass ? j.ghist : nullptr;
        max_tiles = std::max(max_tiles, t.ntiles);
        ++nt;
    }
    if (nt == 0) return HSK_OK;
    a.ntask = nt; a.k = c->cfg.kmer_size; a.npass = npass;
    if (npass) memcpy(a.pass, plan, sizeof(PassDesc) * npass);
    const size_t dyn = (size_t)std::max(npass, 1) * 256 * 4;
    // persistent workgroups: exactly what is resident at once (a second wave would start when the first is done)
    static std::map<size_t, int> occ_c[2];               // per dynamic-LDS size (the histogram area grows with the pass count)
    int &occ = occ_c[ext ? 1 : 0][dyn];
    if (!occ) {
        int nb = 0;
        hipError_t e = ext ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, expand_kernel<NW, true>, EXP_THREADS, dyn)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, expand_kernel<NW, false>, EXP_THREADS, dyn);
        occ = (e == hipSuccess && nb > 0) ? nb : 4;
    }
    u32 rw = (u32)std::max(1, occ * 256 / (8 * nt));
    rw = (u32)std::min<u64>(rw, (max_tiles + 7) / 8);
    a.row_workers = std::max<u32>(rw, 1);
    a.nrows = max_tiles;
    const u32 grid = 8u * (u32)nt * a.row_workers;
    if (ext) hipLaunchKernelGGL((expand_kernel<NW, true>), dim3(grid), dim3(EXP_THREADS), dyn, stream, a);
    else hipLaunchKernelGGL((expand_kernel<NW, false>), dim3(grid), dim3(EXP_THREADS), dyn, stream, a);
    HIPCHK(c, hipGetLastError());
    if (!pre) { for (int i = 0; i < nt; ++i) expand_release(c, x[i]); c->pool.release(d_kcur); }
    return HSK_OK;
}

template <int NW>
static int expand_task(hsk_ctx *c, const TaskSegs &ts, const u8 *sm_len, const BaseSource &src, const u32 *sm_pos, const int32_t *sm_rid,
                       u64 *d_keys, u64 *d_vals)
{
    ExpandJob j; j.ts = &ts; j.sm_len = sm_len; j.src = src; j.sm_pos = sm_pos; j.sm_rid = sm_rid; j.keys = d_keys; j.vals = d_vals; j.ghist = nullptr;
    return expand_batch<NW>(c, &j, 1);
}

// multi-GPU: bytes of all supermers of the store, in storage order (what the exchange sends)
// run_kernel = false: only the byte array is allocated; the bytes of a task group are produced right before the group
// travels (pack_group, on the communication stream, overlapped with the sort of the previous group)
static int pack_store_bytes(hsk_ctx *c, SupermerStore &st, const BaseSource &src, bool run_kernel = true)
{
    if (st.bytes_done) { st.base = src; st.group_packed.assign(4096, 1); return HSK_OK; }      // byte-store mode: the placement already wrote them
    DALLOC(c, st.sm_bytes, u8 *, st.tot_bytes + 64);
    st.base = src;
    if (st.tot_sup == 0 || !run_kernel) return HSK_OK;
    TaskSegs all; ExpSeg s; s.sup_off = 0; s.n_sup = st.tot_sup; s.byte_off = 0; s.kmer_off = 0; s.tile_start = 0;
    all.segs.push_back(s);
    all.ntiles = (st.tot_sup + EXP_TILE - 1) / EXP_TILE;
    ExpandScratch x;
    int rc = expand_prepare(c, all, st.sm_len, x); if (rc) return rc;
    hipLaunchKernelGGL(pack_kernel, dim3((u32)all.ntiles), dim3(EXP_THREADS), 0, c->stream, x.d_segs, 1, st.sm_len, src.src8, src.bit0, src.nwords,
                       st.sm_gpos, x.d_tile_off, st.sm_bytes);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hsk_sync(c, c->stream));          // `all` lives on this stack frame
    expand_release(c, x);
    return HSK_OK;
}

// The bytes of the supermer ranges one task group sends (one contiguous range per destination rank), packed on `stream`.
// The scratch of the job must have been allocated (pack_group_alloc) before the caller fenced `stream` against the main
// stream; the job (host-side segment list included) stays alive until the group has been released.
struct PackJob { TaskSegs ts; ExpandScratch x; };

static int pack_group_alloc(hsk_ctx *c, const ExchangePlan &sp, int nranks, PackJob &job)
{
    job = PackJob();
    u64 tile = 0;
    for (int q = 0; q < nranks; ++q) {
        if (!sp.send_sup[q]) continue;
        ExpSeg sg; sg.sup_off = sp.send_sup_off[q]; sg.n_sup = sp.send_sup[q]; sg.byte_off = sp.send_byte_off[q]; sg.kmer_off = 0; sg.tile_start = tile;
        tile += (sg.n_sup + EXP_TILE - 1) / EXP_TILE;
        job.ts.segs.push_back(sg);
    }
    job.ts.ntiles = tile;
    if (!tile) return HSK_OK;
    DALLOC(c, job.x.d_segs, ExpSeg *, sizeof(ExpSeg) * job.ts.segs.size());
    DALLOC(c, job.x.d_tile_sum, u64 *, tile * 16 + 64);
    DALLOC(c, job.x.d_tile_off, u64 *, tile * 16 + 64);
    return HSK_OK;
}

static int pack_group_launch(hsk_ctx *c, const SupermerStore &st, PackJob &job, hipStream_t stream)
{
    if (!job.ts.ntiles) return HSK_OK;
    const TaskSegs *tp = &job.ts; const u8 *lens = st.sm_len;
    int rc = expand_prepare_batch(c, 1, &tp, &lens, &job.x, stream, true); if (rc) return rc;
    hipLaunchKernelGGL(pack_kernel, dim3((u32)job.ts.ntiles), dim3(EXP_THREADS), 0, stream, job.x.d_segs, (int)job.ts.segs.size(), st.sm_len,
                       st.base.src8, st.base.bit0, st.base.nwords, st.sm_gpos, job.x.d_tile_off, st.sm_bytes);
    HIPCHK(c, hipGetLastError());
    return HSK_OK;
}
