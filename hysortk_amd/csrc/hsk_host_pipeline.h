// hsk_host_pipeline.h -- the whole path: exchange feeder, heavy-hitter lists, per-rank task loop, single-GPU / RCCL / virtual-rank drivers.
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// ------------------------------------------------------------------------------------------------
// Exchange / sort overlap (multi-GPU).  The owned tasks of every rank are cut into groups of
// XCD_BATCH consecutive tasks; group g+1 travels on `comm_stream` (RCCL send/recv, or device copies
// between the virtual ranks of the loopback driver) while group g is expanded, sorted and counted on
// the main stream.  The reference overlaps the same way with BATCH-sized MPI_Ialltoallv rounds
// (src/kmerops.cpp:130-196, exchange_supermer's stage loop); here the unit is a task group so that a
// sort batch never waits for bytes it does not need.  HSK_OVERLAP=0 selects one exchange up front.
// ------------------------------------------------------------------------------------------------
static int estimate_plan(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads, int nranks);      // hsk_api.hip
static u32 certain_drop_mask(hsk_ctx *c);                                                                                                          // hsk_api.hip

struct TaskInput { const u8 *len; BaseSource src; const u32 *pos; const int32_t *rid; const unsigned short *sub16 = nullptr; };

// HSK_TEST_FAIL="<rank>:<site>" (tests/test_gpu_rccl.py, read at every call): the named step of that rank fails as if an
// allocation had returned null.  Sites: sortbuf (before the first task group travels), group1 (the exchange buffers of the
// second group: peers are already inside the exchange), late (the second batch: everything is in flight).
static bool test_fail(hsk_ctx *c, const char *site)
{
    const char *e = getenv("HSK_TEST_FAIL");
    if (!e || !*e) return false;
    const char *colon = strchr(e, ':');
    return colon && atoi(e) == c->comm.rank && strcmp(colon + 1, site) == 0;
}

struct GroupFeeder {
    hsk_ctx *c = nullptr;
    int nranks = 1, rank = 0, ngroups = 0;
    bool ext = false;
    std::vector<int32_t> group_of;                     // task -> group inside its owner's task list
    std::vector<ExchangePlan> pl;                      // [group] this rank's plan
    std::vector<ExchangeBuffers> xb;                   // [group] receive arrays, alive from post to release
    std::vector<hipEvent_t> arrived;                   // [group] recorded on comm_stream after the transfer
    int posted = 0, released = 0;
    // transport: RCCL (store of this rank) or loopback (stores and plans of all virtual ranks)
    SupermerStore *st = nullptr;
    std::vector<SupermerStore> *st_all = nullptr;
    std::vector<std::vector<PackJob>> packs;           // [group] byte-packing jobs issued with the group (scratch released with it)
    bool lazy_pack = false;                            // the stores' bytes are produced group by group (pack_group_*)
    bool live = false;                                 // RCCL: every rank got past the last agreement before the exchange and will post every group
    bool with_sub = false;                             // every rank's store carries the supermers' minimizer bits (sm_sub16): they travel, the owners take the combining extraction
    const std::vector<std::vector<ExchangePlan>> *pl_all = nullptr;     // [rank][group]
    u64 bytes_moved = 0;

    int plan(hsk_ctx *c_, int nranks_, int rank_, u32 ntasks, const std::vector<int32_t> &owner, const std::vector<u32> &order,
             const std::vector<u64> &M, const std::vector<u64> &task_base, std::vector<TaskSegs> &segs)
    {
        c = c_; nranks = nranks_; rank = rank_; ext = c->cfg.extension != 0;
        assign_task_groups(nranks, ntasks, owner, XCD_BATCH, group_of, ngroups);
        pl.resize(ngroups); xb.resize(ngroups); arrived.assign(ngroups, nullptr); packs.assign(ngroups, std::vector<PackJob>());
        segs.assign(ntasks, TaskSegs());
        for (int g = 0; g < ngroups; ++g) plan_exchange(nranks, rank, ntasks, owner, order, M, task_base, pl[g], segs, &group_of, g);
        return HSK_OK;
    }
    int post(int g)
    {
        ExchangeBuffers &b = xb[g]; const ExchangePlan &p = pl[g];
        if (g == 1 && !draining && test_fail(c, "group1")) return fail(c, HSK_ERR_OOM, "exchange buffers of group %d (injected)", g);
        b.len = (u8 *)c->pool.alloc(p.recv_tot_sup + 64); b.bytes = (u8 *)c->pool.alloc(p.recv_tot_bytes + 64); b.nbytes = p.recv_tot_bytes;
        if (ext) { b.pos = (u32 *)c->pool.alloc(p.recv_tot_sup * 4 + 64); b.rid = (int32_t *)c->pool.alloc(p.recv_tot_sup * 4 + 64); }
        if (with_sub) b.sub16 = (unsigned short *)c->pool.alloc(p.recv_tot_sup * 2 + 64);
        if (!b.len || !b.bytes || (ext && (!b.pos || !b.rid)) || (with_sub && !b.sub16)) return fail(c, HSK_ERR_OOM, "exchange buffers of group %d", g);
        // the bytes this group sends are packed now, on the communication stream (RCCL: this rank's store; virtual ranks:
        // the store of every source that has not produced group g yet)
        std::vector<SupermerStore *> to_pack;
        if (lazy_pack) {
            if (st_all) { for (int src = 0; src < nranks; ++src) { SupermerStore &ss = (*st_all)[src]; if (ss.group_packed.size() <= (size_t)g) ss.group_packed.resize(ngroups, 0); if (!ss.group_packed[g]) to_pack.push_back(&ss); } }
            else { if (st->group_packed.size() <= (size_t)g) st->group_packed.resize(ngroups, 0); if (!st->group_packed[g]) to_pack.push_back(st); }
            packs[g].resize(to_pack.size());
            for (size_t i = 0; i < to_pack.size(); ++i) {
                const ExchangePlan &sp = st_all ? (*pl_all)[(int)(to_pack[i] - &(*st_all)[0])][g] : p;
                int rc = pack_group_alloc(c, sp, nranks, packs[g][i]); if (rc) return rc;
            }
        }
        // the pool hands out blocks whose previous user may still be running on the main stream: order the
        // transfer after everything launched there so far (that is the work of group g-2 and earlier)
        hipEvent_t fence = ev_get(c);
        HIPCHK(c, hipEventRecord(fence, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->comm_stream, fence, 0));
        ev_put(c, fence);
        hipStream_t s = c->comm_stream;
        for (size_t i = 0; i < to_pack.size(); ++i) { int rc = pack_group_launch(c, *to_pack[i], packs[g][i], s); if (rc) return rc; to_pack[i]->group_packed[g] = 1; }
        if (st_all) {
            for (int src = 0; src < nranks; ++src) {
                const ExchangePlan &sp = (*pl_all)[src][g]; const SupermerStore &ss = (*st_all)[src];
                const u64 n = sp.send_sup[rank], nb = sp.send_bytes[rank];
                if (n != p.recv_sup[src] || nb != p.recv_bytes[src]) return fail(c, HSK_ERR_INTERNAL, "exchange plan mismatch %d->%d (group %d)", src, rank, g);
                if (!n) continue;
                HIPCHK(c, hipMemcpyAsync(b.len + p.recv_sup_off[src], ss.sm_len + sp.send_sup_off[rank], n, hipMemcpyDeviceToDevice, s));
                HIPCHK(c, hipMemcpyAsync(b.bytes + p.recv_byte_off[src], ss.sm_bytes + sp.send_byte_off[rank], nb, hipMemcpyDeviceToDevice, s));
                if (with_sub) HIPCHK(c, hipMemcpyAsync(b.sub16 + p.recv_sup_off[src], ss.sm_sub16 + sp.send_sup_off[rank], n * 2, hipMemcpyDeviceToDevice, s));
                if (ext) {
                    HIPCHK(c, hipMemcpyAsync(b.pos + p.recv_sup_off[src], ss.sm_pos + sp.send_sup_off[rank], n * 4, hipMemcpyDeviceToDevice, s));
                    HIPCHK(c, hipMemcpyAsync(b.rid + p.recv_sup_off[src], ss.sm_rid + sp.send_sup_off[rank], n * 4, hipMemcpyDeviceToDevice, s));
                }
            }
        } else {
            int rc = post_exchange(c->comm, s, ext, p, st->sm_len, st->sm_bytes, st->sm_pos, st->sm_rid, b, with_sub ? st->sm_sub16 : nullptr);
            if (rc) return fail(c, HSK_ERR_COMM, "supermer exchange (group %d) failed: %d (%s)", g, rc, c->comm.last_error.c_str());
        }
        bytes_moved += p.recv_tot_bytes + p.recv_tot_sup * (ext ? 9 : 1) + (with_sub ? p.recv_tot_sup * 2 : 0);
        arrived[g] = ev_get(c);
        HIPCHK(c, hipEventRecord(arrived[g], s));
        return HSK_OK;
    }
    // the main stream is about to read group g: make sure g and g+1 are on their way, wait for g
    int need(int g)
    {
        const int upto = std::min(g + 1, ngroups - 1);
        while (posted <= upto) { int rc = post(posted); if (rc) return rc; ++posted; }
        HIPCHK(c, hipStreamWaitEvent(c->stream, arrived[g], 0));
        return HSK_OK;
    }
    // the main stream has launched its last reader of every group below g
    void release_below(int g)
    {
        for (; released < g && released < posted; ++released) {
            xb[released].release(c->pool);         // next user is ordered after the readers by post()'s fence (or is on the main stream)
            for (auto &pj : packs[released]) expand_release(c, pj.x);
            packs[released].clear();
            if (arrived[released]) { ev_put(c, arrived[released]); arrived[released] = nullptr; }
        }
    }
    // This rank's count has failed after the exchange began (RCCL only).  Its peers still expect its supermers and still send
    // it theirs: a rank that simply returned would leave them blocked in ncclRecv for ever (the reference dies together there:
    // MPI_Abort, src/kmerops.cpp:1477).  So the rank drains its own work, hands every device block the failed count allocated
    // (`keep`: the live blocks before it, i.e. the supermer store stays) back to the pool -- a failed allocation is the usual
    // reason to be here -- and posts the remaining groups one by one, receiving into buffers it drops at once.  The ranks then
    // agree on the outcome (run_pipeline) and all return an error.
    bool draining = false;
    int drain_after_failure(const std::vector<void *> &keep)
    {
        draining = true;
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamSynchronize(c->d2h_stream);
        for (int g = 0; g < ngroups; ++g) { xb[g] = ExchangeBuffers(); packs[g].clear(); if (arrived[g]) { ev_put(c, arrived[g]); arrived[g] = nullptr; } }
        released = posted;
        c->pool.release_all_but(keep);
        for (; posted < ngroups; ++posted) {
            const int g = posted;
            int rc = post(g); if (rc) return rc;
            HIPCHK(c, hipStreamSynchronize(c->comm_stream));
            xb[g].release(c->pool);
            for (auto &pj : packs[g]) expand_release(c, pj.x);
            packs[g].clear();
            if (arrived[g]) { ev_put(c, arrived[g]); arrived[g] = nullptr; }
            released = g + 1;
        }
        return HSK_OK;
    }
    // every rank must take part in every group even when it owns no task of it
    int finish()
    {
        while (posted < ngroups) { int rc = post(posted); if (rc) return rc; ++posted; }
        HIPCHK(c, hsk_sync(c, c->comm_stream));
        release_below(ngroups);
        return HSK_OK;
    }
    TaskInput input(u32 t) const
    {
        const ExchangeBuffers &b = xb[group_of[t]];
        TaskInput in; in.len = b.len; in.src = source_from_bytes(b.bytes, b.nbytes); in.pos = b.pos; in.rid = b.rid; in.sub16 = b.sub16;
        return in;
    }
};

static bool overlap_enabled()
{
    return tune("overlap", 1) != 0;
}

// ---- heavy-hitter tasks (a8): the owner's side --------------------------------------------------------------
// d_entries: the {k-mer, count} lists of all ranks for one task, concatenated (n entries, each list key-ordered,
// a key at most once per list).  Orders them by key with the count as payload, sums equal keys, filters [L, U].
template <int NW>
static int heavy_merge_task(hsk_ctx *c, const u64 *d_entries, u64 n, u64 *d_histo, u32 histo_len, TaskOut &out)
{
    out = TaskOut();
    if (n == 0) return HSK_OK;
    u64 *kA, *kB, *vA, *vB;
    DALLOC(c, kA, u64 *, n * NW * 8 + 64); DALLOC(c, kB, u64 *, n * NW * 8 + 64); DALLOC(c, vA, u64 *, n * 8 + 64); DALLOC(c, vB, u64 *, n * 8 + 64);
    hipLaunchKernelGGL(heavy_split_kernel<NW>, dim3((u32)std::min<u64>((n + HV_THREADS - 1) / HV_THREADS, 4096)), dim3(HV_THREADS), 0, c->stream, d_entries, n, kA, vA);
    SortScratch sc; int rc = alloc_sort_scratch(c, sc); if (rc) return rc;
    u64 *sk, *sv;
    rc = sort_task_device<NW>(c, kA, kB, vA, vB, n, c->cfg.kmer_size, sc, &sk, &sv);
    free_sort_scratch(c, sc);
    if (rc) return rc;
    rc = merge_sorted_pairs<NW>(c, sk, sv, n, d_histo, histo_len, out); if (rc) return rc;
    HIPCHK(c, hsk_sync(c, c->stream));
    c->pool.release(kA); c->pool.release(kB); c->pool.release(vA); c->pool.release(vB);
    return HSK_OK;
}

// Host threads that widen compact result batches (pack_entries_kernel's k-mer words + 16-bit counts, copied into pinned staging)
// into the caller-visible entries while the GPU counts the next batches.  Every thread of a batch waits for the batch's copy
// event, then takes its slice.  The destructor joins: no thread outlives the call that started it.
struct WidenPiece {                                                 // one task's share of a batch
    hipEvent_t copied; const u64 *keys; const unsigned short *cnts; u64 *dst; u64 n;
    // prefix form (one-word keys): low 48 key bits as u32 + u16, counts of cw bytes, dir[p] = first entry of prefix p (dir[65536] = n)
    const u32 *lo32 = nullptr; const unsigned short *mid16 = nullptr; const u8 *cnt8 = nullptr; const u32 *dir = nullptr; int cw = 0;
};
struct WidenPool {
    hsk_ctx *c;
    std::vector<std::thread> th;
    std::vector<hipEvent_t> evs;
    explicit WidenPool(hsk_ctx *c_) : c(c_) {}
    static int nthreads()
    {
        { const int v = (int)tune("widen_threads", 0); if (v > 0) return std::min(v, 64); }
        static const int n = []() { const unsigned hc = std::thread::hardware_concurrency(); return (int)std::min<unsigned>(32, std::max<unsigned>(2, hc / 2)); }();
        return n;
    }
    // A batch arrives task by task (one copy + one event per piece): thread t widens slice t of every piece in turn, so that all
    // threads are done shortly after the LAST piece has landed -- the tail of the call is one piece's widening, not one batch's.
    void add(const std::vector<WidenPiece> &pieces, int nw)
    {
        for (auto &p : pieces) evs.push_back(p.copied);
        const int nt = nthreads(), dev = c->cfg.device;
        for (int t = 0; t < nt; ++t) {
            th.emplace_back([=]() {
                (void)hipSetDevice(dev);
                for (const WidenPiece &p : pieces) {
                    (void)hipEventSynchronize(p.copied);
                    const u64 lo = p.n * (u64)t / nt, hi = p.n * (u64)(t + 1) / nt;
                    const u64 *keys = p.keys; const unsigned short *cnts = p.cnts; u64 *dst = p.dst;
                    if (p.dir) {                                       // prefix form: the top 16 key bits come from the directory
                        if (lo >= hi) continue;
                        typedef unsigned long long v2u64p __attribute__((vector_size(16)));
                        u32 pl = 0, ph = 65536;                        // last prefix that starts at or before entry lo
                        while (ph - pl > 1) { const u32 mid = (pl + ph) >> 1; if ((u64)p.dir[mid] <= lo) pl = mid; else ph = mid; }
                        u32 pre = pl; u64 next = p.dir[pre + 1];
                        const bool nt = ((uintptr_t)dst & 15) == 0;
                        for (u64 i = lo; i < hi; ++i) {
                            while (i >= next) { ++pre; next = p.dir[pre + 1]; }
                            const unsigned long long key = ((unsigned long long)pre << 48) | ((unsigned long long)p.mid16[i] << 32) | p.lo32[i];
                            const unsigned long long cv = p.cw == 1 ? (unsigned long long)p.cnt8[i] : (unsigned long long)reinterpret_cast<const unsigned short *>(p.cnt8)[i];
                            if (nt) { const v2u64p e = {key, cv}; __builtin_nontemporal_store(e, (v2u64p *)dst + i); }
                            else { dst[2 * i] = key; dst[2 * i + 1] = cv; }
                        }
                        continue;
                    }
                    // one-word keys: an entry is one aligned 16-byte store that nobody reads back soon -- non-temporal (no read for
                    // ownership: a plain store loop is bound by the cache lines it first has to fetch)
                    typedef unsigned long long v2u64 __attribute__((vector_size(16)));
                    if (nw == 1 && ((uintptr_t)dst & 15) == 0) for (u64 i = lo; i < hi; ++i) { const v2u64 e = {keys[i], (unsigned long long)cnts[i]}; __builtin_nontemporal_store(e, (v2u64 *)dst + i); }
                    else if (nw == 1) for (u64 i = lo; i < hi; ++i) { dst[2 * i] = keys[i]; dst[2 * i + 1] = cnts[i]; }
                    else for (u64 i = lo; i < hi; ++i) { for (int w = 0; w < nw; ++w) dst[i * (nw + 1) + w] = keys[i * nw + w]; dst[i * (nw + 1) + nw] = cnts[i]; }
                }
            });
        }
    }
    void join() { for (auto &t : th) if (t.joinable()) t.join(); th.clear(); for (auto e : evs) ev_put(c, e); evs.clear(); }
    ~WidenPool() { join(); }
};

struct HeavyIn { u32 task; u64 *d_entries; u64 n; };       // a heavy task this rank owns: concatenated lists of all ranks
struct ProcExtra {
    bool force_batch = false;                              // every task through the batch kernels (partial batches padded)
    const std::vector<HeavyIn> *heavy_in = nullptr;        // merged and filtered here (they have no supermers)
    u32 vt_shift = 0;                                      // item-mode store: minimizer bits the parse's virtual tasks have consumed (a segment's virtual task: ExpSeg::byte_off)
    SupermerStore *items_store = nullptr;                  // item-mode store of one GPU: its items go back to the pool as soon as the bucket order has read them
};

// Everything after the supermers of the owned tasks are in place: per task expand, sort, count; then the
// result of this rank.  `segs[t]` lists where the supermers of task t live (x_len / x_src / x_pos / x_rid).
template <int NW>
static int process_rank(hsk_ctx *c, u32 ntasks, const std::vector<int32_t> &owner, int rank, std::vector<TaskSegs> &segs,
                        const u8 *x_len, const BaseSource &x_src, const u32 *x_pos, const int32_t *x_rid,
                        hsk_result *out, ResultPriv *rp, PhaseTimer &pt, bool pt_total_open, GroupFeeder *feeder = nullptr,
                        const ProcExtra *ex = nullptr)
{
    const bool ext = c->cfg.extension != 0;
    const int K = c->cfg.kmer_size;
    u64 max_task = 0, total_kmers = 0;
    for (u32 t = 0; t < ntasks; ++t) { finalize_segs(segs[t]); max_task = std::max(max_task, segs[t].nkmers); total_kmers += segs[t].nkmers; }
    out->total_kmers = total_kmers;
    if (c->est.valid && NW == 1 && !ext && !c->forbid_long_way && max_task) {
        // distinct keys per 16-bit prefix bin of the largest task, from this call's estimate: the first table of the ladder, or no tables at all
        // (most bins beyond 2048 slots: four prefix passes + the tile finish; what a batch used to find out the hard way, agg_stage2)
        const double d = c->est.distinct_per_kmer * (double)max_task / 65536.0;
        c->agg_first_cap = d <= 600.0 ? AG_LOG2CAP_SMALL : d <= 1250.0 ? AG_LOG2CAP_MEDIUM : AG_LOG2CAP_LARGE;
        if (d > 1450.0 && !(ex && ex->vt_shift) && !x_src.item) { c->agg_off = true; c->agg_off_calls = 0; }
    }

    // ---- per task: expand, sort, count ---------------------------------------------------------------
    const u32 histo_len = (u32)std::min<int64_t>((int64_t)c->cfg.upper_freq + 1, 65536);    // (U <= 65535 except in the unfiltered pre-aggregation)
    u64 *d_histo; DALLOC(c, d_histo, u64 *, (size_t)histo_len * 8);
    HIPCHK(c, hipMemsetAsync(d_histo, 0, (size_t)histo_len * 8, c->stream));
    // Tasks are sorted eight at a time, one per XCD (sort_batch_device); a remainder of fewer than eight
    // tasks goes through the single-task kernel.  HSK_XCD_BATCH=0 forces the single-task path.
    const bool batch_env = tune("xcd_batch", 1) != 0;
    const bool batch_enabled = batch_env && c->xcd_batch_ok;          // the one-task-per-XCD kernels need all eight XCDs (hsk_init's census)
    std::vector<u32> mine;
    for (u32 t = 0; t < ntasks; ++t) if (owner[t] == rank && segs[t].nkmers) mine.push_back(t);
    // A remainder of three or more tasks is padded to a full batch with empty slots (an XCD without a task idles, which
    // still beats eight full-width passes per task on the single-task path); ex->force_batch pads any remainder.
    const u32 EMPTY_TASK = ~0u;
    TaskSegs empty_segs;
    std::vector<TaskOut> touts(ntasks);
    // several ranks, the supermers arrived with their minimizer bits: the owner builds the items, batch by batch (hsk_combine.h, 1b)
    const bool fed_wanted = NW <= 2 && feeder && feeder->with_sub && c->combine_now && !ext;
    const bool forced = ((ex && ex->force_batch) || x_src.item != nullptr || fed_wanted) && batch_enabled;      // (item-mode store: every task goes through whole batches)
    // a caller's task count below eight (the reference's default for one rank is five): three to seven tasks of some size still
    // go faster as one padded batch (5/8 of the batch path's rate) than one by one on the single-task path (about 1/3 of it)
    u64 mine_kmers = 0; for (u32 t : mine) mine_kmers += segs[t].nkmers;
    const bool small_batch = batch_enabled && mine.size() >= 3 && mine.size() < (size_t)XCD_BATCH && mine_kmers >= (1ULL << 25);
    if ((batch_enabled && mine.size() >= (size_t)XCD_BATCH && mine.size() % XCD_BATCH >= 3) || (forced && !mine.empty()) || small_batch)
        while (mine.size() % XCD_BATCH) mine.push_back(EMPTY_TASK);
    const bool batch = batch_enabled && mine.size() >= (size_t)XCD_BATCH;
    const int nsets = batch ? XCD_BATCH : 1;
    // fused finish: one-word keys (aggregating or tile finish), two-word keys with K >= 40 (aggregating finish only)
    const bool fused = !ext && finish_enabled() && (NW == 1 ? hybrid_enabled() : (NW <= 3 && agg_enabled() && prefix_plan_ok<NW>(K, true)));
    const bool agg = fused && agg_enabled();
    // EXTENSION with one-word keys: two passes on the top 16 bits (payload carried) + grouping aggregation
    const bool fused_ext = ext && NW <= 3 && hybrid_enabled() && finish_enabled() && agg_enabled() && prefix_plan_ok<NW>(K, true);
    // Two batches in flight on ONE stream (two sets of sort buffers): the host enqueues expand / scatter / aggregation of
    // batch b + 1 BEFORE it waits for the aggregation totals of batch b, sizes batch b's outputs and enqueues its
    // compaction.  The GPU therefore never runs dry while the host waits (HSK_LAG=0: one batch at a time, every wait drains
    // the stream).  An earlier version expanded batch b + 1 on a second stream beside the sort of batch b (HSK_PIPELINE):
    // every kernel of the path already fills the chip, the gain was 1 %, and it is gone.
    // Measured (10 Gbp, 5 batches): the batch interval is the same with and without the lag (18.7 / 18.8 ms: a drained
    // stream costs well under 0.1 ms against ~19 ms of kernels per batch), but the lag delays every batch's compaction,
    // and with it the batch's result copy, by one batch: host results take 214 ms with it and 197 ms without.  Default: on
    // when the result stays in HBM (nothing to copy), off when it goes to the host; HSK_LAG=0/1 forces it.
    const bool keep_dev = (c->cfg.flags & HSK_FLAG_KEEP_DEVICE) != 0;
    const int lag_env = (int)tune("lag", -1);
    const bool lag_enabled = lag_env < 0 ? keep_dev : lag_env != 0;
    const bool lag = batch && agg && NW <= 2 && lag_enabled && mine.size() >= 2 * (size_t)XCD_BATCH;
    const int nslot = lag ? 2 : 1;
    // expand fused with the first scatter pass (hsk_scatter.h): one-word keys, aggregating finish, whole batches
    // (EXTENSION: payload chunks beside the key chunks, HSK_FUSED_SCATTER_EXT=0 turns that variant off)
    const bool xs_ext_enabled = tune("fused_scatter_ext", 1) != 0;
    const bool xs_wide_enabled = tune("fused_scatter_wide", 1) != 0;      // two-word keys
    constexpr int XS_CH = XsCfg<(NW <= 2 ? NW : 1)>::CHUNK;
    const bool xs = batch && (NW == 1 ? (!ext || xs_ext_enabled) : (NW == 2 && !ext && xs_wide_enabled && prefix_top_bits(K, NW) == 16)) && scatter_enabled() &&
                    scatter_store_keys(max_task, XS_CH) < (1ULL << 32) && finish_enabled() && hybrid_enabled() && agg_enabled() && prefix_plan_ok<NW>(K, true);
    // the combining extraction (hsk_combine.h): the store carries the supermers' minimizer bits, whole batches, the aggregating finish
    bool combine = false;
    // (an item-mode store -- x_src.item -- holds nothing the instance path could read: every batch takes the combining extraction, or the
    //  call starts again without it)
    const bool item_mode = x_src.item != nullptr;
    bool fed_combine = false;
    if constexpr (NW <= 2) {
        combine = item_mode && xs && agg && !(NW == 1 ? c->agg_off : c->agg_off_wide) && !ext && !feeder && mine.size() % XCD_BATCH == 0;
        fed_combine = fed_wanted && !item_mode && xs && agg && !(NW == 1 ? c->agg_off : c->agg_off_wide) && batch && mine.size() % XCD_BATCH == 0;
        combine = combine || fed_combine;
    }
    if (item_mode && !combine) { c->combine_veto = true; return retry_plan("an item-mode store, but no batch to combine (xs / agg / batch)", (xs ? 1u : 0u) | (agg ? 2u : 0u) | (batch ? 4u : 0u) | ((mine.size() % XCD_BATCH == 0) ? 8u : 0u)); }
    BucketOrder border;
    BucketOrder border_fed[2]; FedItems fed_items[2];      // per slot: the items and the bucket order of the batch in flight (several ranks)
    bool slot_combine[2] = {false, false};
    ScatterBatch sbatch[2];                               // per slot
    PassDesc xs_plan[MAX_PASSES];
    u64 *kAs[2][XCD_BATCH] = {{nullptr}}, *kBs[2][XCD_BATCH] = {{nullptr}}, *vAs[2][XCD_BATCH] = {{nullptr}}, *vBs[2][XCD_BATCH] = {{nullptr}};
    u64 **kA = kAs[0], **kB = kBs[0], **vA = vAs[0], **vB = vBs[0];          // slot 0: also the single-task path
    SortScratch sc;
    u64 *d_ghist_slot[2] = {nullptr, nullptr};
    // The combining extraction's buffers hold {k-mer, count} PAIRS, not k-mers: with the call's own estimate of the input (estimate_plan: distinct k-mers
    // per k-mer) they are sized for four times the pairs it promises (+ 2 % of the k-mers) instead of one record per k-mer -- 20 GB instead of 102 at
    // 10 Gbp, and device memory is what a process's FIRST call pays for (~20-60 ms per GB mapped for the first time, tools/exp/malloc_cost.hip).  A call
    // whose pairs do not fit after all (error bit 512: the last chunk takes what runs over) starts again with full-sized buffers.  Several ranks: full
    // size (nobody starts again while peers wait).
    u64 rec_cap = max_task;
    if (combine && !fed_combine && c->est.valid && !c->pair_cap_full && tune("pair_cap", 1) != 0)
        rec_cap = std::min<u64>(max_task, std::max<u64>((u64)((double)max_task * std::min(1.0, 4.0 * c->est.distinct_per_kmer * c->est_bias + 0.02)), 1ULL << 22));
    if (tune("pair_cap_records", 0) > 0 && combine && !fed_combine) rec_cap = std::min<u64>(max_task, (u64)tune("pair_cap_records", 0));      // (tests: stores that run over)
    auto alloc_sort_buffers = [&]() -> int {
        if (max_task) {
            for (int sl = 0; sl < nslot; ++sl) for (int i = 0; i < nsets; ++i) {
                DALLOC(c, kAs[sl][i], u64 *, rec_cap * NW * 8 + 64);
                DALLOC(c, kBs[sl][i], u64 *, (xs ? scatter_store_keys(rec_cap, XS_CH) + XS_CH : rec_cap) * NW * 8 + 64);   // xs: the chunk store of the first pass (+ the chunk that takes what runs over)
                if (ext || combine) { DALLOC(c, vAs[sl][i], u64 *, rec_cap * 8 + 64); DALLOC(c, vBs[sl][i], u64 *, (xs ? scatter_store_keys(rec_cap, XS_CH) + XS_CH : rec_cap) * 8 + 64); }
            }
            int rc = alloc_sort_scratch(c, sc); if (rc) return rc;
        }
        if (batch) for (int sl = 0; sl < nslot; ++sl) DALLOC(c, d_ghist_slot[sl], u64 *, (size_t)XCD_BATCH * MAX_PASSES * 256 * 8);
        return HSK_OK;
    };
    // The bucket order of ALL owned tasks comes before the sort buffers are allocated: once its scatter is enqueued nobody reads the item store
    // again, and its 21 GB (10 Gbp) go back to the pool: the call's peak of live device memory 68 -> 46 GB (HSK_TIMING prints the pool's state;
    // what the pool has MAPPED stays at 89 GB -- it hands a cached block only to requests of nearly its size -- and that, mapped for the first
    // time, is what a process's first call pays for).
    if (combine && !fed_combine) {
        pt.begin(PH_EXTRACT);
        int rc = bucket_order_tasks(c, ntasks, segs, mine, x_src, ex ? ex->vt_shift : 0, border); if (rc) return rc;
        pt.end(PH_EXTRACT);
        if (!border.active) { c->combine_veto = true; return retry_plan("no bucket order"); }
        if (ex && ex->items_store) {                      // (stream-ordered reuse: every later user of these blocks is enqueued behind the scatter)
            SupermerStore &is = *ex->items_store;
            c->pool.release(is.sm_item); is.sm_item = nullptr; c->pool.release(is.sm_sub); is.sm_sub = nullptr;
            c->pool.release(is.d_bitems); is.d_bitems = nullptr; is.n_bitems = 0; for (void *&q : is.bin_aux) { c->pool.release(q); q = nullptr; }
        }
    }
    {
        int arc = alloc_sort_buffers();
        if (!arc && feeder && test_fail(c, "sortbuf")) arc = fail(c, HSK_ERR_OOM, "sort buffers (injected)");
        if (feeder && !feeder->st_all && c->comm.active()) {
            // the largest allocations of the call are behind us: make sure EVERY rank got them before the first task group
            // travels (a rank that gave up here alone would leave its peers blocked in their first send / receive)
            std::vector<u64> none;
            const int st_ = c->comm.allreduce_with_status(none, RCCL_MAX, arc != 0, c->stream, c->pool);
            if (st_ < 0) return fail(c, HSK_ERR_COMM, "allreduce(status) failed: %d (%s)", st_, c->comm.last_error.c_str());
            if (st_ > 0) return arc ? arc : fail(c, HSK_ERR_COMM, "another rank ran out of memory before the supermer exchange");
            feeder->live = true;                          // from here on a failing rank drains the exchange and the ranks agree at the end (run_pipeline)
        } else if (arc) return arc;
    }
    u64 n_total = 0, pay_total = 0;
    // payload offsets are global over the owned tasks in ascending id: prefix of k-mer counts
    std::vector<u64> pay_before(ntasks, 0);
    { u64 acc = 0; for (u32 t : mine) { if (t == EMPTY_TASK) continue; pay_before[t] = acc; if (ext) acc += segs[t].nkmers; } }
    TaskInput dflt; dflt.len = x_len; dflt.src = x_src; dflt.pos = x_pos; dflt.rid = x_rid;
    // ---- result copies that overlap the kernels (host result, no payload, tasks finished in ascending id) ------------
    // The pinned block is sized from the entries-per-k-mer ratio of the previous call (or of this call's first batch);
    // should the list outgrow it, the early copies are abandoned and everything is copied again at the end.
    const bool keep = keep_dev;
    const bool early_enabled = tune("early_d2h", 1) != 0;
    bool early = batch && agg && NW <= 2 && !keep && !ext && early_enabled && !(ex && ex->heavy_in && !ex->heavy_in->empty());
    u64 *early_buf = nullptr; u64 early_cap = 0, early_used = 0, early_kmers = 0;
    // compact copies (HSK_COMPACT_D2H=0: entries travel as they are): counts fit 16 bits whenever the filter's upper bound does
    // HSK_COMPACT_D2H: 0 entries as they are (16 bytes), 1 k-mer words + 16-bit counts (10 bytes), 2 (default) the prefix form for
    // one-word keys (7 bytes with U <= 255, else 8; + 256 KB of directory per task)
    const int compact_mode = (int)tune("compact_d2h", 2);
    const bool compact = compact_mode > 0 && c->cfg.upper_freq <= 65535;
    WidenPool widen(c);
    u64 compact_bytes = 0, compact_entries = 0;
    std::vector<void *> pk_dev, pk_host;                  // device / pinned staging of the compact batches (handed back when the call ends)
    std::vector<u8> copied(ntasks, 0);
    std::vector<EvPair> d2h_ev;
    const bool profile_ev = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    auto early_copy = [&](const u32 *tasks, int ntk) -> int {
        if (!early) return HSK_OK;
        u64 nb = 0, kb = 0;
        for (int i = 0; i < ntk; ++i) if (tasks[i] != EMPTY_TASK) { nb += touts[tasks[i]].n; kb += segs[tasks[i]].nkmers; }
        if (!early_buf) {
            const double ratio = c->entries_per_kmer > 0 ? c->entries_per_kmer : (kb ? (double)nb / (double)kb : 1.0);
            early_cap = (u64)(ratio * 1.08 * (double)total_kmers) + (1u << 16);
            if (early_cap * (NW + 1) * 8 > (64ULL << 30)) { early = false; return HSK_OK; }      // not worth pinning that much on a guess
            early_buf = (u64 *)host_alloc(c, rp, early_cap * (NW + 1) * 8);
            if (!early_buf) { early = false; return HSK_OK; }
        }
        if (early_used + nb > early_cap) { early = false; return HSK_OK; }                       // the guess was too small: copy at the end
        // compact: every task's entries are packed on the main stream ([k-mer words][16-bit counts], 16-byte aligned per task), copied
        // task by task into pinned staging and widened into early_buf by host threads while the next batch is counted
        u8 *d_pk = nullptr, *h_pk = nullptr;
        size_t pk_off[XCD_BATCH + 1] = {0}, pk_len[XCD_BATCH] = {0};
        // one-word keys: the prefix form (7 or 8 bytes per entry + a 256 KB directory per task); otherwise k-mer words + 16-bit counts
        const bool prefix_form = NW == 1 && compact_mode >= 2;
        const int cw = c->cfg.upper_freq <= 255 ? 1 : 2;
        constexpr size_t DIR_BYTES = (size_t)65537 * 4 + 12;            // (padded to 16 bytes)
        if (compact && nb) {
            bool fits = true;
            for (int i = 0; i < ntk; ++i) {
                const u64 n_i = (tasks[i] == EMPTY_TASK) ? 0 : touts[tasks[i]].n;
                if (n_i >= 0xFFFFFFF0ULL) fits = false;
                pk_len[i] = prefix_form ? (n_i ? (((size_t)n_i * 4 + 15) & ~(size_t)15) + (((size_t)n_i * 2 + 15) & ~(size_t)15) + (((size_t)n_i * cw + 15) & ~(size_t)15) + DIR_BYTES : 0)
                                        : (size_t)n_i * (NW * 8 + 2);
                pk_off[i + 1] = pk_off[i] + ((pk_len[i] + 15) & ~(size_t)15);
            }
            const size_t pk_bytes = pk_off[ntk] + 64;
            if (fits) { d_pk = (u8 *)c->pool.alloc(pk_bytes); h_pk = (u8 *)host_alloc(c, rp, pk_bytes); }
            if (!d_pk || !h_pk) { c->pool.release(d_pk); if (h_pk) host_release(c, rp, h_pk); d_pk = nullptr; h_pk = nullptr; }      // (no room: this batch travels as it is)
            else {
                pk_dev.push_back(d_pk); pk_host.push_back(h_pk);
                for (int i = 0; i < ntk; ++i) {
                    if (tasks[i] == EMPTY_TASK || !touts[tasks[i]].n) continue;
                    const TaskOut &to = touts[tasks[i]];
                    const u32 grid = (u32)std::min<u64>((to.n + 255) / 256, 2048);
                    if (prefix_form) {
                        if constexpr (NW == 1) {
                            u8 *b = d_pk + pk_off[i];
                            u32 *lo32 = (u32 *)b; unsigned short *mid16 = (unsigned short *)(b + (((size_t)to.n * 4 + 15) & ~(size_t)15));
                            u8 *cnt = (u8 *)mid16 + (((size_t)to.n * 2 + 15) & ~(size_t)15);
                            u32 *dir = (u32 *)(cnt + (((size_t)to.n * cw + 15) & ~(size_t)15));
                            HIPCHK(c, hipMemsetAsync(dir, 0xFF, (size_t)65537 * 4, c->stream));
                            if (cw == 1) hipLaunchKernelGGL((pack_entries_prefix_kernel<u8>), dim3(grid), dim3(256), 0, c->stream, to.entries, to.n, lo32, mid16, cnt, dir);
                            else hipLaunchKernelGGL((pack_entries_prefix_kernel<unsigned short>), dim3(grid), dim3(256), 0, c->stream, to.entries, to.n, lo32, mid16, (unsigned short *)cnt, dir);
                            hipLaunchKernelGGL(pack_dir_close_kernel, dim3(1), dim3(1024), 0, c->stream, dir, (u32)to.n);
                        }
                    } else hipLaunchKernelGGL(pack_entries_kernel, dim3(grid), dim3(256), 0, c->stream, to.entries, to.n, NW,
                                              (u64 *)(d_pk + pk_off[i]), (unsigned short *)(d_pk + pk_off[i] + (size_t)to.n * NW * 8));
                }
            }
        }
        hipEvent_t done = ev_get(c);
        HIPCHK(c, hipEventRecord(done, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->d2h_stream, done, 0));
        ev_put(c, done);
        EvPair ep{}; if (profile_ev) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 6; (void)hipEventRecord(ep.a, c->d2h_stream); }
        if (d_pk) {
            std::vector<WidenPiece> pieces;
            u64 o = 0;
            for (int i = 0; i < ntk; ++i) {
                if (tasks[i] == EMPTY_TASK || !touts[tasks[i]].n) continue;
                const u64 n_i = touts[tasks[i]].n;
                HIPCHK(c, hipMemcpyAsync(h_pk + pk_off[i], d_pk + pk_off[i], pk_len[i], hipMemcpyDeviceToHost, c->d2h_stream));
                WidenPiece wp; wp.copied = ev_get(c);
                HIPCHK(c, hipEventRecord(wp.copied, c->d2h_stream));
                wp.keys = nullptr; wp.cnts = nullptr;
                if (prefix_form) {
                    const u8 *b = h_pk + pk_off[i];
                    wp.lo32 = (const u32 *)b; wp.mid16 = (const unsigned short *)(b + (((size_t)n_i * 4 + 15) & ~(size_t)15));
                    wp.cnt8 = (const u8 *)wp.mid16 + (((size_t)n_i * 2 + 15) & ~(size_t)15);
                    wp.dir = (const u32 *)(wp.cnt8 + (((size_t)n_i * cw + 15) & ~(size_t)15)); wp.cw = cw;
                } else { wp.keys = (const u64 *)(h_pk + pk_off[i]); wp.cnts = (const unsigned short *)(h_pk + pk_off[i] + (size_t)n_i * NW * 8); }
                wp.dst = early_buf + (early_used + o) * (NW + 1); wp.n = n_i;
                pieces.push_back(wp);
                o += n_i;
                compact_bytes += pk_len[i];
            }
            widen.add(pieces, NW);
            compact_entries += nb;
        }
        for (int i = 0; i < ntk; ++i) {
            const u32 t = tasks[i];
            if (t == EMPTY_TASK) continue;
            TaskOut &to = touts[t];
            if (to.n && !d_pk) HIPCHK(c, hipMemcpyAsync(early_buf + early_used * (NW + 1), to.entries, to.n * (NW + 1) * 8, hipMemcpyDeviceToHost, c->d2h_stream));
            early_used += to.n; copied[t] = 1;
        }
        if (profile_ev) { (void)hipEventRecord(ep.b, c->d2h_stream); d2h_ev.push_back(ep); }
        early_kmers += kb;
        return HSK_OK;
    };
    int slot_prefix[2] = {0, 0};                          // the digit plan a slot's batch was expanded for
    // ... and which finish follows it: the aggregation (slot_agg), the grouping aggregation of EXTENSION (slot_fext), or -- once
    // hsk_ctx::agg_off / agg_off_wide have found the input to hold (nearly) only unique k-mers, possibly in the middle of a call --
    // the tile finish (one-word keys) / full-width passes and the two-pass counter (everything else); slot_follow: a finish that
    // wants prefix passes only
    bool slot_agg[2] = {false, false}, slot_fext[2] = {false, false}, slot_follow[2] = {false, false};
    BatchTask bts[2][XCD_BATCH];
    // one launch expands the eight tasks mine[bpos ..] into the slot's buffers and counts the digits of the passes that follow
    auto issue_expand = [&](size_t bpos, int sl) -> int {
        slot_agg[sl] = agg && !(NW == 1 ? c->agg_off : c->agg_off_wide);
        slot_fext[sl] = fused_ext && !c->agg_off_wide;
        slot_follow[sl] = slot_agg[sl] || slot_fext[sl] || (NW == 1 && fused);
        const bool will_combine = combine && (fed_combine || border.active) && slot_agg[sl];
        if (combine && !will_combine && !fed_combine) { c->combine_veto = true; return retry_plan("a batch that cannot take the combining extraction"); }      // (several ranks: the batch simply takes the instance path -- nobody starts a call again while peers wait)
        // (two-word keys: the finish orders a bin's keys by the bits below a 16-bit prefix, agg_order_many -- their pairs take bins of 16 bits)
        const int prefix_bits = will_combine ? (NW == 1 ? combine_prefix_bits(c) : AG_PREFIX_BITS) : (slot_agg[sl] || slot_fext[sl]) ? AG_PREFIX_BITS : 64 - HYBRID_SHIFT;
        slot_prefix[sl] = prefix_bits;
        PassDesc plan[MAX_PASSES];
        int npass = batch_pass_plan<NW>(c, K, slot_follow[sl], will_combine ? AG_PREFIX_BITS : prefix_bits, plan);
        if (will_combine) { npass = 2; plan[0] = PassDesc{NW - 1, 64 - prefix_bits, prefix_bits - 8}; plan[1] = PassDesc{NW - 1, 56, 8}; }      // the pairs' two digits (most significant word): the low prefix bits, then the top 8
        pt.begin(PH_EXTRACT);
        HIPCHK(c, hipMemsetAsync(d_ghist_slot[sl], 0, (size_t)XCD_BATCH * MAX_PASSES * 256 * 8, c->stream));
        ExpandJob jobs[XCD_BATCH];
        for (int i = 0; i < XCD_BATCH; ++i) {
            const u32 t = mine[bpos + i];
            BatchTask &b = bts[sl][i];
            b = BatchTask();
            b.kA = kAs[sl][i]; b.kB = kBs[sl][i];
            if (ext || will_combine) { b.vA = vAs[sl][i]; b.vB = vBs[sl][i]; }      // (the payload buffers mean "records carry a payload" to everything downstream)
            if (t == EMPTY_TASK) { jobs[i] = ExpandJob(); jobs[i].ts = &empty_segs; continue; }
            b.n = segs[t].nkmers;
            const TaskInput in = feeder ? feeder->input(t) : dflt;
            jobs[i].ts = &segs[t]; jobs[i].sm_len = in.len; jobs[i].src = in.src; jobs[i].sm_pos = in.pos; jobs[i].sm_rid = in.rid;
            jobs[i].keys = b.kA; jobs[i].vals = b.vA; jobs[i].ghist = d_ghist_slot[sl] + (size_t)i * MAX_PASSES * 256;
        }
        int rc;
        slot_combine[sl] = false;
        if (will_combine) {
            if constexpr (NW <= 2) {
                u32 tk[XCD_BATCH]; u64 *gh[XCD_BATCH];
                for (int i = 0; i < XCD_BATCH; ++i) { tk[i] = mine[bpos + i]; gh[i] = d_ghist_slot[sl] + (size_t)i * MAX_PASSES * 256; }
                memcpy(xs_plan, plan, sizeof(PassDesc) * 2);
                u64 *h_nout = (u64 *)((char *)c->pinned + c->pinned_bytes - 2048 + (size_t)sl * 128);
                if (fed_combine) {
                    const unsigned short *s16[XCD_BATCH];
                    for (int i = 0; i < XCD_BATCH; ++i) s16[i] = tk[i] == EMPTY_TASK ? nullptr : feeder->input(tk[i]).sub16;
                    std::vector<TaskSegs> gsegs; BaseSource gsrc;
                    rc = build_items_batch(c, ntasks, tk, jobs, s16, gsegs, gsrc, fed_items[sl], c->stream); if (rc) return rc;
                    rc = bucket_order_tasks(c, ntasks, gsegs, std::vector<u32>(tk, tk + XCD_BATCH), gsrc, FED_VT_SHIFT, border_fed[sl]); if (rc) return rc;
                    if (!border_fed[sl].active) return fail(c, HSK_ERR_UNSUPPORTED, "a task of 2^32 supermers and more");
                    rc = combine_batch<NW>(c, tk, bts[sl], gh, plan, border_fed[sl], h_nout, sbatch[sl], c->stream, rec_cap); if (rc) return rc;
                    fed_release(c, fed_items[sl]); bucket_release(c, border_fed[sl]);      // (stream-ordered reuse: their readers are enqueued)
                } else { rc = combine_batch<NW>(c, tk, bts[sl], gh, plan, border, h_nout, sbatch[sl], c->stream, rec_cap); if (rc) return rc; }
                slot_combine[sl] = sbatch[sl].active;
            }
        } else
        if (xs && (slot_agg[sl] || slot_fext[sl]) && npass == 2 && plan[0].bits == 8 && plan[1].bits == 8) {
            if constexpr (NW <= 2) {
                for (int i = 0; i < XCD_BATCH; ++i) { jobs[i].keys = bts[sl][i].kB; jobs[i].vals = bts[sl][i].vB; }
                memcpy(xs_plan, plan, sizeof(PassDesc) * 2);
                rc = scatter_expand_batch<NW>(c, jobs, bts[sl], plan, sbatch[sl], c->stream); if (rc) return rc;
            }
        } else { rc = expand_batch<NW>(c, jobs, XCD_BATCH, npass, plan, c->stream, nullptr); if (rc) return rc; }
        pt.end(PH_EXTRACT);
        return HSK_OK;
    };
    AggPending pend[2]; size_t pend_pos[2] = {0, 0};
    // second stage of the aggregating finish of the batch in slot sl: totals -> outputs -> compaction -> result copy
    auto finish_stage2 = [&](int sl, bool covered) -> int {
        if constexpr (NW <= 3) {
            TaskOut fo[XCD_BATCH];
            pt.begin(PH_COUNT);
            int rc = agg_stage2<NW>(c, pend[sl], d_histo, histo_len, fo, covered);
            pt.end(PH_COUNT);
            tmark("batch stage 2 done (totals waited for, compaction enqueued)");
            if (rc) return rc;
            u32 tk[XCD_BATCH];
            for (int i = 0; i < XCD_BATCH; ++i) { tk[i] = mine[pend_pos[sl] + i]; if (tk[i] != EMPTY_TASK) touts[tk[i]] = fo[i]; }
            for (int i = 0; i < XCD_BATCH; ++i) if (tk[i] != EMPTY_TASK && fo[i].failed) {
                early = false;
                // a bin of pairs beyond the last table of the weighted finish: this call again, on the instance path (dispatch_pipeline)
                if (combine) {
                    if (!combine_prefix_forced() && slot_prefix[sl] < COMBINE_PREFIX_MAX) c->combine_prefix = c->combine_prefix_floor = COMBINE_PREFIX_MAX;      // once more with the narrowest bins
                    else if (!c->combine_off) c->leave_combine();
                    c->distrust_estimate((double)combine_ratio());
                    return retry_plan("a bin beyond the weighted finish");
                }
            }
            return early_copy(tk, XCD_BATCH);
        }
        return HSK_OK;
    };
    u64 comb_pairs = 0, comb_kmers = 0;                   // pairs the combining extraction has written / k-mers they stand for (this call)
    size_t pos = 0;
    const size_t nbatch = batch ? mine.size() / XCD_BATCH : 0;
    for (size_t b = 0; b < nbatch; ++b, pos += XCD_BATCH) {
        const int sl = lag ? (int)(b & 1) : 0;
        if (b == 1 && feeder && test_fail(c, "late")) return fail(c, HSK_ERR_OOM, "second batch (injected)");
        if (pend[sl].active) { int rc = finish_stage2(sl, false); if (rc) return rc; }       // (only after hsk_ctx::agg_off ended the aggregation in the middle of the call: the slot's buffers are about to be reused)
        if (feeder) {                                   // exposed (not overlapped) part of the exchange
            pt.begin(PH_EXCH);
            for (int i = 0; i < XCD_BATCH; ++i) { if (mine[pos + i] == EMPTY_TASK) continue; int rc = feeder->need(feeder->group_of[mine[pos + i]]); if (rc) return rc; }
            pt.end(PH_EXCH);
        }
        { int rc = issue_expand(pos, sl); if (rc) return rc; }
        BatchTask *bt = bts[sl];
        if (feeder) feeder->release_below((pos + XCD_BATCH < mine.size() && mine[pos + XCD_BATCH] != EMPTY_TASK) ? feeder->group_of[mine[pos + XCD_BATCH]] : feeder->ngroups);
        const int prefix_bits = slot_prefix[sl];
        if (slot_combine[sl]) {
            // the pairs of every task are known on the device only: one wait per batch (the kernels behind it are sized from the answer)
            c->stats.host_syncs++;
            HIPCHK(c, hsk_sync(c, c->stream));
            const u64 *h_nout = (const u64 *)((char *)c->pinned + c->pinned_bytes - 2048 + (size_t)sl * 128);
            if ((u32)h_nout[XCD_BATCH] & 512u) {                   // the pair stores ran over (they were sized from the estimate): once more, sized for the k-mers
                (void)hipMemsetAsync(c->d_err, 0, 4, c->stream);
                if (fed_combine) return fail(c, HSK_ERR_INTERNAL, "pair stores overrun with several ranks");
                c->pair_cap_full = true;
                return retry_plan("the pair stores ran over (sized from the estimate)");
            }
            u64 bp = 0, bk = 0, pmax = 0;
            for (int i = 0; i < XCD_BATCH; ++i) { if (mine[pos + i] == EMPTY_TASK || !bt[i].n) continue; bk += bt[i].n; bt[i].n = h_nout[i]; bp += h_nout[i]; pmax = std::max<u64>(pmax, h_nout[i]); }
            c->combine_prefix = std::max(c->combine_prefix_floor, combine_prefix_for(pmax));      // (the batches and calls after this one)
            comb_pairs += bp; comb_kmers += bk; c->stats.combine_pairs += bp;
            if (timing_enabled()) fprintf(stderr, "[hsk] combining extraction: %llu pairs for %llu k-mers\n", (unsigned long long)bp, (unsigned long long)bk);
            // More than one pair per sixteen k-mers: this input has too few copies per k-mer for the detour to pay (measured on 10 Gbp, DESIGN.md 3.2d:
            // one pair per 25.6 k-mers 102 against 126 ms, one per 6.6 -- reads with 0.3 % errors -- 171 against 138: the tables overflow inside
            // the buckets and the parse side's extra 20 ms buy nothing; at one per sixteen a bucket's table is already 37 % full); the batches of this call finish on the pairs, the next calls take the
            // instance path
            const u64 ratio_env = combine_ratio();
            if (bk && bp * ratio_env > bk && !c->combine_off) {
                c->leave_combine();
                if (c->est.valid) c->est_bias = std::min(8.0, std::max(1.0, ((double)bp / (double)bk) / c->est.distinct_per_kmer));      // (the estimate promised fewer pairs: later estimates on this context are scaled)
            }
        }
        pt.begin(PH_SORT);
        if (sbatch[sl].active) { if constexpr (NW <= 2) { int rc = sort_batch_prescattered<NW>(c, bt, xs_plan, d_ghist_slot[sl], sbatch[sl]); if (rc) return rc; } }
        else { int rc = sort_batch_device<NW>(c, bt, K, slot_follow[sl], prefix_bits, d_ghist_slot[sl]); if (rc) return rc; }
        pt.end(PH_SORT);
        if (slot_fext[sl]) {
            if constexpr (NW <= 3) {
                pt.begin(PH_COUNT);
                TaskOut fo[XCD_BATCH]; u64 pb[XCD_BATCH];
                for (int i = 0; i < XCD_BATCH; ++i) pb[i] = mine[pos + i] != EMPTY_TASK ? pay_before[mine[pos + i]] : 0;
                int rc = agg_ext_finish_batch_device<NW>(c, bt, K, pb, d_histo, histo_len, fo); if (rc) return rc;
                for (int i = 0; i < XCD_BATCH; ++i) if (mine[pos + i] != EMPTY_TASK) touts[mine[pos + i]] = fo[i];
                pt.end(PH_COUNT);
            }
        } else if (agg && slot_agg[sl]) {
            if constexpr (NW <= 3) {
                // the previous batch first: this batch's expand and scatter pass are queued behind its aggregation, so the
                // wait for its totals does not idle the GPU, and its compaction (and result copy) starts one kernel earlier
                // (stage 2 of the previous batch BEFORE this batch's stage 1 would start its result copy one kernel earlier, but a
                // device-to-host copy running beside agg_finish_kernel stretches a batch from 19 to 32 ms: measured in round 2, gone)
                pt.begin(PH_COUNT);
                int rc = agg_stage1<NW>(c, bt, K, prefix_bits, sl, pend[sl], slot_combine[sl]);
                pt.end(PH_COUNT);
                if (rc) return rc;
                pend_pos[sl] = pos;
                if (lag && b > 0 && pend[sl ^ 1].active) { rc = finish_stage2(sl ^ 1, true); if (rc) return rc; }
                if (!lag) { rc = finish_stage2(sl, false); if (rc) return rc; }
            }
        } else if (fused && NW == 1 && !ext) {
            if constexpr (NW == 1) {
                pt.begin(PH_COUNT);
                TaskOut fo[XCD_BATCH];
                int rc = finish_batch_device<1>(c, bt, K, max_task, d_histo, histo_len, fo); if (rc) return rc;
                for (int i = 0; i < XCD_BATCH; ++i) if (mine[pos + i] != EMPTY_TASK) touts[mine[pos + i]] = fo[i];
                pt.end(PH_COUNT);
            }
        } else {
            pt.begin(PH_COUNT);
            for (int i = 0; i < XCD_BATCH; ++i) {
                const u32 t = mine[pos + i];
                if (t == EMPTY_TASK) continue;
                int rc = count_task_device<NW>(c, bt[i].out_k, bt[i].out_v, bt[i].n, pay_before[t], d_histo, histo_len, touts[t]); if (rc) return rc;
            }
            pt.end(PH_COUNT);
        }
    }
    if (lag && agg && nbatch > 0) for (int sl = 0; sl < 2; ++sl) if (pend[sl].active) { int rc = finish_stage2(sl, false); if (rc) return rc; }
    for (; pos < mine.size(); ++pos) {
        const u32 t = mine[pos];
        const u64 n = segs[t].nkmers;
        int rc;
        if (feeder) { pt.begin(PH_EXCH); rc = feeder->need(feeder->group_of[t]); pt.end(PH_EXCH); if (rc) return rc; }
        pt.begin(PH_EXTRACT);
        const TaskInput in = feeder ? feeder->input(t) : dflt;
        rc = expand_task<NW>(c, segs[t], in.len, in.src, in.pos, in.rid, kA[0], ext ? vA[0] : nullptr); if (rc) return rc;
        pt.end(PH_EXTRACT);
        if (feeder) feeder->release_below(pos + 1 < mine.size() ? feeder->group_of[mine[pos + 1]] : feeder->ngroups);
        pt.begin(PH_SORT);
        u64 *sk, *sv;
        rc = sort_task_device<NW>(c, kA[0], kB[0], ext ? vA[0] : nullptr, ext ? vB[0] : nullptr, n, K, sc, &sk, &sv); if (rc) return rc;
        pt.end(PH_SORT);
        pt.begin(PH_COUNT);
        rc = count_task_device<NW>(c, sk, sv, n, pay_before[t], d_histo, histo_len, touts[t]); if (rc) return rc;
        pt.end(PH_COUNT);
    }
    // heavy-hitter tasks this rank owns arrive as k-mer lists: order, sum, filter
    if (ex && ex->heavy_in) {
        pt.begin(PH_COUNT);
        for (const HeavyIn &hv : *ex->heavy_in) {
            { int rc = heavy_merge_task<NW>(c, hv.d_entries, hv.n, d_histo, histo_len, touts[hv.task]); if (rc) return rc; }
            mine.push_back(hv.task);
        }
        pt.end(PH_COUNT);
    }
    for (u32 t : mine) { if (t == EMPTY_TASK) continue; touts[t].pay_base = pay_before[t]; n_total += touts[t].n; pay_total += touts[t].npay; }
    if (feeder) { int rc = feeder->finish(); if (rc) return rc; }
    for (int sl = 0; sl < nslot; ++sl) for (int i = 0; i < nsets; ++i) { c->pool.release(kAs[sl][i]); c->pool.release(kBs[sl][i]); c->pool.release(vAs[sl][i]); c->pool.release(vBs[sl][i]); }
    free_sort_scratch(c, sc);
    if (combine && !c->combine_off) c->combine_good_calls++;
    bucket_release(c, border);
    c->pool.release(d_ghist_slot[0]); c->pool.release(d_ghist_slot[1]);

    // ---- result ----------------------------------------------------------------------------------------
    pt.begin(PH_D2H);
    out->n = n_total;
    out->task_off = (uint64_t *)host_alloc(c, rp, (size_t)(ntasks + 1) * 8);
    out->histo = (uint64_t *)host_alloc(c, rp, (size_t)histo_len * 8);
    out->histo_len = histo_len;
    if (!out->task_off || !out->histo) return fail(c, HSK_ERR_OOM, "pinned host allocation failed");
    HIPCHK(c, hipMemcpyAsync(out->histo, d_histo, (size_t)histo_len * 8, hipMemcpyDeviceToHost, c->stream));
    u32 *h_err = (u32 *)((char *)c->pinned + c->pinned_bytes - 64);        // the sticky device error word travels with the result
    HIPCHK(c, hipMemcpyAsync(h_err, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
    if (!keep) {
        // the early copies are good if they stayed inside the block and cover a prefix of the list (tasks in ascending id)
        bool early_ok = early_buf != nullptr && early && n_total <= early_cap;
        if (early_ok) { bool gap = false; for (u32 t = 0; t < ntasks && early_ok; ++t) { if (!touts[t].n) continue; if (!copied[t]) gap = true; else if (gap) early_ok = false; } }
        if (early_buf && !early_ok) { HIPCHK(c, hsk_sync(c, c->d2h_stream)); widen.join(); host_release(c, rp, early_buf); early_buf = nullptr; std::fill(copied.begin(), copied.end(), 0); compact_bytes = compact_entries = 0; }
        out->entries = early_buf ? early_buf : (uint64_t *)host_alloc(c, rp, n_total * (NW + 1) * 8);
        if (!out->entries) return fail(c, HSK_ERR_OOM, "pinned host allocation of %llu bytes failed", (unsigned long long)(n_total * (NW + 1) * 8));
        if (ext) {
            out->payload_off = (uint64_t *)host_alloc(c, rp, (n_total + 1) * 8);
            out->pos = (uint32_t *)host_alloc(c, rp, pay_total * 4);
            out->rid = (int32_t *)host_alloc(c, rp, pay_total * 4);
            if (!out->payload_off || !out->pos || !out->rid) return fail(c, HSK_ERR_OOM, "pinned host allocation failed");
        }
    }
    EvPair d2h_tail{}; if (profile_ev && !keep) { d2h_tail.a = ev_get(c); d2h_tail.b = ev_get(c); d2h_tail.kind = 6; (void)hipEventRecord(d2h_tail.a, c->stream); }
    u64 o = 0, po = 0;
    for (u32 t = 0; t < ntasks; ++t) {
        out->task_off[t] = o;
        TaskOut &to = touts[t];
        if (!keep) {
            if (to.n && !copied[t]) { HIPCHK(c, hipMemcpyAsync(out->entries + o * (NW + 1), to.entries, to.n * (NW + 1) * 8, hipMemcpyDeviceToHost, c->stream)); c->stats.d2h_bytes += 0; }
            if (ext && to.n) HIPCHK(c, hipMemcpyAsync(out->payload_off + o, to.payoff, to.n * 8, hipMemcpyDeviceToHost, c->stream));
            if (ext && to.npay) {
                HIPCHK(c, hipMemcpyAsync(out->pos + po, to.pos, to.npay * 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(out->rid + po, to.rid, to.npay * 4, hipMemcpyDeviceToHost, c->stream));
            }
        }
        o += to.n; po += to.npay;
    }
    out->task_off[ntasks] = o;
    if (profile_ev && !keep) { (void)hipEventRecord(d2h_tail.b, c->stream); d2h_ev.push_back(d2h_tail); }
    if (!keep) c->stats.d2h_bytes += (n_total - compact_entries) * (NW + 1) * 8 + compact_bytes + (ext ? (n_total + 1) * 8 + pay_total * 8 : 0);
    pt.end(PH_D2H);
    if (pt_total_open) pt.end(PH_TOTAL);
    tmark("result copies enqueued");
    HIPCHK(c, hsk_sync(c, c->stream));
    tmark("main stream drained");
    if (early_buf) HIPCHK(c, hsk_sync(c, c->d2h_stream));
    tmark("copy stream drained");
    widen.join();                                         // the last batch's entries are being widened
    for (void *p : pk_dev) c->pool.release(p);
    for (void *p : pk_host) host_release(c, rp, p);
    tmark("entries widened");
    for (auto &e : d2h_ev) c->ev_pending.push_back(e);
    if (*h_err) {
        const u32 w = *h_err;
        (void)hipMemsetAsync(c->d_err, 0, 4, c->stream);
        return fail(c, HSK_ERR_INTERNAL, "device-side check failed (error word %u:%s%s%s%s%s)", w, (w & 1) ? " radix look-back timed out;" : "",
                    (w & 2) ? " chunk map wait timed out;" : "", (w & 4) ? " foreign supermer;" : "",
                    (w & 8) ? " an XCD did not expand its task (cursors / histogram do not add up);" : "", (w & 16) ? " an XCD did not drain its sort task;" : "");
    }
    if (!ext && total_kmers && !c->forbid_long_way) c->entries_per_kmer = (double)n_total / (double)total_kmers;
    if (ext && !keep) out->payload_off[n_total] = pay_total;
    if (keep) { rp->dev_tasks = touts; out->entries_dev = nullptr; }
    else for (auto &to : touts) free_task_out(c, to);
    c->pool.release(d_histo);
    if (pt_total_open) out->ms_total = pt.collect(PH_TOTAL);
    out->ms_parse = pt.collect(PH_PARSE); out->ms_exchange = pt.collect(PH_EXCH);
    out->ms_extract = pt.collect(PH_EXTRACT); out->ms_sort = pt.collect(PH_SORT); out->ms_count = pt.collect(PH_COUNT);
    out->ms_d2h = pt.collect(PH_D2H);
    return HSK_OK;
}

// ---- heavy-hitter tasks (a8): the sending side ----------------------------------------------------------------
// HeavyHitterClassifier (reference src/kmerops.cpp:1157-1199) on the GLOBAL k-mer counts, for every key width (the
// reference's ScatteredKmerList is generic over TKmer, kmerops.cpp:363-401); forced plain with EXTENSION or
// PLAIN_CLASSIFIER (kmerops.cpp:109-113).
static bool heavy_enabled(hsk_ctx *c, int /*nw*/, int nranks)
{
    return nranks > 1 && c->cfg.extension == 0 && (c->cfg.flags & HSK_FLAG_PLAIN_CLASSIFIER) == 0;
}

// Every rank turns its OWN supermers of the heavy tasks into unfiltered {k-mer, count} lists (ScatteredKmerList,
// kmerops.cpp:363-398): only the heavy tasks are placed, then the ordinary expand / sort / aggregate kernels run with
// L = 1, U = max.  lists[t] stays in HBM; failed[t] = the aggregating finish could not handle the task (it is then
// sent as supermers like any other task -- on every rank, the flags are combined by the caller).
template <int NW>
static int heavy_preaggregate(hsk_ctx *c, ParseJob &job, const u8 *d_packed, u64 packed_bytes, const std::vector<u8> &is_heavy,
                              std::vector<TaskOut> &lists, std::vector<u8> &failed)
{
    const u32 ntasks = job.ntasks;
    lists.assign(ntasks, TaskOut()); failed.assign(ntasks, 0);
    std::vector<u32> order; std::vector<u8> skip(ntasks, 0);
    for (u32 t = 0; t < ntasks; ++t) if (is_heavy[t]) order.push_back(t);
    for (u32 t = 0; t < ntasks; ++t) if (!is_heavy[t]) { order.push_back(t); skip[t] = 1; }
    SupermerStore sth;
    int rc = parse_place(c, job, order, sth, &skip); if (rc) return rc;
    std::vector<TaskSegs> segs(ntasks);
    std::vector<int32_t> own(ntasks, -1);
    for (u32 t = 0; t < ntasks; ++t) {
        if (!is_heavy[t]) continue;
        own[t] = 0;
        if (sth.task_tot[3 * t] == 0) continue;
        ExpSeg sg; sg.sup_off = sth.task_base[3 * t]; sg.n_sup = sth.task_tot[3 * t]; sg.byte_off = sth.task_base[3 * t + 1]; sg.kmer_off = 0; sg.tile_start = 0;
        segs[t].segs.push_back(sg); segs[t].nkmers = sth.task_tot[3 * t + 2];
    }
    const hsk_config keep = c->cfg;
    c->cfg.lower_freq = 1; c->cfg.upper_freq = INT32_MAX; c->cfg.flags |= HSK_FLAG_KEEP_DEVICE;
    c->forbid_long_way = true;
    hsk_result tmp; memset(&tmp, 0, sizeof tmp);
    ResultPriv *rp = new ResultPriv(); tmp.priv = rp; tmp.nw = NW;
    PhaseTimer pt(c);
    ProcExtra ex; ex.force_batch = true;
    rc = process_rank<NW>(c, ntasks, own, 0, segs, sth.sm_len, source_from_store(sth, d_packed, packed_bytes), nullptr, nullptr, &tmp, rp, pt, false, nullptr, &ex);
    c->cfg = keep; c->forbid_long_way = false;
    if (rc == HSK_OK) {
        for (u32 t = 0; t < ntasks; ++t) if (is_heavy[t] && t < rp->dev_tasks.size()) { lists[t] = rp->dev_tasks[t]; failed[t] = lists[t].failed ? 1 : 0; }
        rp->dev_tasks.clear();                              // the lists are ours now
    }
    hsk_result_free(c, &tmp);
    free_store(c, sth);
    return rc;
}

template <int NW>
static int run_pipeline(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                        int64_t rid_base, hsk_result *out)
{
    const bool ext = c->cfg.extension != 0;
    const int K = c->cfg.kmer_size;
    const int nranks = c->comm.active() ? c->comm.nranks : 1;
    const int rank = c->comm.active() ? c->comm.rank : 0;
    memset(out, 0, sizeof *out);
    ResultPriv *rp = new ResultPriv();
    out->priv = rp; out->nw = NW;
    if ((c->agg_off || c->agg_off_wide) && ++c->agg_off_calls >= 8) { c->agg_off = c->agg_off_wide = false; c->agg_off_calls = 0; }      // (another look every eighth call: the input may have changed)
    if (c->combine_off && ++c->combine_off_calls >= c->combine_off_period) { c->combine_off = false; c->combine_off_calls = 0; c->combine_prefix_floor = 0; }      // (another look: the bin width starts from the default again as well)
    // the combining extraction pays from a few hundred million k-mers on (a bucket order of the supermers comes first); HSK_COMBINE_MIN_BYTES
    // moves the limit (tests: 0)
    const u64 combine_min = (u64)tune("combine_min_bytes", 64LL << 20);
    // this call's own estimate of the input (estimate_plan) decides where there is one; the context's memory of earlier calls (combine_off, agg_off) where there is none
    const bool est = c->est.valid;
    const bool combine_pays = !c->combine_left_now && (est ? c->est.distinct_per_kmer * c->est_bias * (double)combine_ratio() <= 1.0 : !c->combine_off);
    if (est && c->agg_off && c->plan_attempt == 0) { c->agg_off = false; c->agg_off_calls = 0; }      // (process_rank decides again, from the estimate and the task sizes)
    // several ranks (round 4): the supermers travel as byte runs with 16 of their minimizer bits, the OWNER of a task builds the items (hsk_combine.h, 1b);
    // needs the grouped exchange and the byte-store placement, and every rank's consent (below)
    // (two-word keys: 40 <= K <= 55 -- the prefix bits sit in the most significant word, an item of 64 bases holds six k-mers and more: shorter
    //  items would be more records per tile than the parse keeps, for 16 bytes that stand for very few k-mers)
    c->combine_now = (NW == 1 || (NW == 2 && K >= 40 && K <= 55)) && !ext && combine_pays && !c->combine_veto && c->plan_attempt < 2 && !(NW == 1 ? c->agg_off : c->agg_off_wide) && combine_enabled() && parse_fast_enabled() &&
                     c->cfg.minimizer_size <= SCAN_MAX_M && packed_bytes >= combine_min && c->xcd_batch_ok &&
                     (nranks == 1 || (overlap_enabled() && place_bytes_enabled(true)));
    c->combine_veto = false;
    // the combining extraction wants buckets of ~12 k k-mers: the parse itself splits every task by the top minimizer bits (virtual
    // tasks, up to 16 per task and HSK_MAX_TASKS in all: ParseArgs::vt_shift), the bucket order does the rest (hsk_combine.h)
    c->vt_shift = 0;
    PhaseTimer pt(c);
    pt.begin(PH_TOTAL);

    u32 ntasks = c->cfg.ntasks ? (u32)c->cfg.ntasks : auto_ntasks(c, packed_bytes, nranks);
    if (c->comm.active()) {
        // every rank must use the same task count (the maximum of the local proposals) and the same plan (the combining extraction only if
        // every rank's own estimate says its input pays for it: the minimizer bits either travel from all ranks or from none)
        // ... and the certain drops: a rank whose own sample holds more than U copies of a homopolymer k-mer is right for all ranks (OR of the masks)
        u64 v[4] = {c->cfg.ntasks ? 0ULL : (u64)ntasks, c->combine_now ? 0ULL : 1ULL, (u64)(c->drop_mask_now & 1u), (u64)((c->drop_mask_now >> 1) & 1u)};
        int rc = c->comm.allreduce_max_u64(v, 4, c->stream, c->pool); if (rc) return fail(c, HSK_ERR_COMM, "allreduce(ntasks, plan) failed: %d", rc);
        if (!c->cfg.ntasks) ntasks = (u32)v[0];
        if (v[1]) c->combine_now = false;
        c->drop_mask_now = (v[2] ? 1u : 0u) | (v[3] ? 2u : 0u);
    }
    out->ntasks = (int32_t)ntasks;
    // (at most 768 virtual tasks: the item placement's LDS holds 16 bytes for each beside its 16384 records; more real tasks than that: the instance path)
    if (ntasks > 768 && nranks == 1) c->combine_now = false;
    if (c->combine_now && nranks == 1) { u32 sh = 0; while (sh < 4 && ((u64)ntasks << (sh + 1)) <= 768) ++sh; c->vt_shift = sh; }
    if (c->combine_now && est && nranks == 1) {
        // a task has at most 2^14 buckets (CS_MAX_LOG2NB; 2^(10 + virtual-task bits)): few, large tasks make buckets whose distinct k-mers overflow the
        // 2048-slot tables again and again (partial pairs: the detour stops paying) -- predicted from the estimate instead of found out by a batch
        const u32 lg = (u32)std::min<int>(CS_MAX_LOG2NB, CS_MAX_LOCAL + (int)c->vt_shift);
        const double per_bucket = (double)packed_bytes * 4.0 / (double)ntasks / (double)(1u << lg);
        if (per_bucket > (double)combine_bucket_kmers() && c->est.distinct_per_kmer * per_bucket > 1400.0) { c->combine_now = false; c->vt_shift = 0; }
    }
    c->item_mode_now = c->combine_now && nranks == 1;      // one GPU: the store holds items (several ranks: byte runs + minimizer bits, the owners build the items)
    const u32 vts = c->vt_shift, nvt = ntasks << vts;       // what the parse calls tasks
    std::vector<int32_t> owner(ntasks, 0);
    std::vector<u32> order(ntasks);
    for (u32 t = 0; t < ntasks; ++t) order[t] = t;

    // ---- parse ------------------------------------------------------------------------------------
    SupermerStore st;
    std::vector<u8> is_heavy(ntasks, 0);
    std::vector<u64> heavy_kmers(ntasks, 0);             // k-mer instances of a heavy task over all ranks (they travel as lists: its owner counts them into total_kmers)
    std::vector<TaskOut> hlists;                         // this rank's {k-mer, count} lists of the heavy tasks
    std::vector<HeavyIn> hin;                            // heavy tasks this rank owns: the lists of all ranks
    bool any_heavy = false;
    int place_rc = HSK_OK;
    std::vector<TaskSegs> segs(ntasks);
    bool pipelined = false;
    pt.begin(PH_PARSE);
    // one GPU, reads arriving from pinned host memory, no payload: ingest, scan and placement as one pipeline over slabs
    const bool pipe_enabled = tune("ingest_pipeline", 1) != 0;
    if (nranks == 1 && !ext && c->h2d_src && pipe_enabled && parse_fast_enabled() && c->cfg.minimizer_size <= SCAN_MAX_M) {
        const u8 *src = c->h2d_src; c->h2d_src = nullptr;
        std::vector<TaskSegs> segs_v;
        const int prc = parse_ingest_pipelined(c, src, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, nvt, st, vts ? segs_v : segs);
        if (prc == HSK_OK) {
            pipelined = true;
            if (vts) {                                          // a task's segments: those of its virtual tasks (which one: ExpSeg::byte_off, unused in item mode)
                for (u32 v = 0; v < nvt; ++v) {
                    TaskSegs &ts = segs[v >> vts];
                    for (ExpSeg sg : segs_v[v].segs) { sg.byte_off = v & ((1u << vts) - 1u); sg.kmer_off = 0; ts.segs.push_back(sg); }
                    ts.nkmers += segs_v[v].nkmers;
                }
            }
        }
        else if (prc == PARSE_FALLBACK && vts) { c->vt_shift = 0; c->combine_veto = true; return retry_plan("the pipelined ingest fell back"); }      // (the fallback parse knows no virtual tasks: the call again, without them)
        else if (prc != PARSE_FALLBACK) { c->vt_shift = 0; return prc; }
        else c->stats.parse_fallbacks++;                        // (the packed reads are in HBM now: the two-step parse below takes it from there)
    }
    if (!pipelined) {
        // the reads are hashed once (parse_count); multi-GPU: the dispatcher needs the global task sizes
        // before the storage order (tasks grouped by owner rank) is known, then parse_place lays the supermers out
        ParseJob job;
        int rc = parse_count(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, nvt, job);
        if (rc && nranks == 1) { parse_release(c, job); c->vt_shift = 0; return rc; }
        if (vts && !job.d_tile_sub && !job.bins.items && !job.empty) { parse_release(c, job); c->vt_shift = 0; c->combine_veto = true; return retry_plan("the parse left its fast path: no items"); }   // (the parse left its fast path: no items)
        if (nranks > 1) {
            // Several ranks: a rank that fails must not return alone (its peers would wait for it in the next collective for
            // ever).  Every all-reduce below carries the ranks' status as one more element; a failed rank keeps taking part
            // (with zeros) until the next one, then all ranks return together.
            Comm &cm = c->comm;
            int local_rc = rc;                                   // first local failure since the last collective
            auto together = [&](int st_, const char *what) -> int {      // result of an all-reduce with status -> return code of this rank
                if (st_ == 0) return HSK_OK;
                if (st_ < 0) return fail(c, HSK_ERR_COMM, "allreduce(%s) failed: %d (%s)", what, st_, cm.last_error.c_str());
                return local_rc ? local_rc : fail(c, HSK_ERR_COMM, "another rank failed before the all-reduce of %s", what);
            };
            std::vector<u64> bytes(ntasks, 0);
            if (!local_rc) for (u32 t = 0; t < ntasks; ++t) bytes[t] = job.task_tot[3 * t + 1] + job.task_tot[3 * t] * (ext ? 9 : 1);
            // heavy-hitter tasks (a8): classified on the global k-mer counts; every rank pre-aggregates its own share
            if (heavy_enabled(c, NW, nranks)) {
                std::vector<u64> kg(ntasks, 0); std::vector<int32_t> types(ntasks, 0);
                if (!local_rc) for (u32 t = 0; t < ntasks; ++t) kg[t] = job.task_tot[3 * t + 2];
                rc = together(cm.allreduce_with_status(kg, RCCL_SUM, local_rc != 0, c->stream, c->pool), "task k-mers");
                if (rc) { parse_release(c, job); return rc; }
                plan_classify(kg.data(), (int)ntasks, c->cfg.unbalanced_ratio, types.data());
                for (u32 t = 0; t < ntasks; ++t) if (types[t] == 1) { is_heavy[t] = 1; any_heavy = true; heavy_kmers[t] = kg[t]; }
            }
            if (any_heavy) {
                std::vector<u8> failed(ntasks, 0);
                local_rc = heavy_preaggregate<NW>(c, job, d_packed, packed_bytes, is_heavy, hlists, failed);
                std::vector<u64> bad(ntasks, 0);
                if (!local_rc) for (u32 t = 0; t < ntasks; ++t) bad[t] = failed[t];
                rc = together(cm.allreduce_with_status(bad, RCCL_MAX, local_rc != 0, c->stream, c->pool), "heavy flags");     // a task one rank could not aggregate travels as supermers everywhere
                if (rc) { parse_release(c, job); return rc; }
                any_heavy = false;
                for (u32 t = 0; t < ntasks; ++t) {
                    if (!is_heavy[t]) continue;
                    if (bad[t]) { is_heavy[t] = 0; free_task_out(c, hlists[t]); continue; }
                    any_heavy = true; c->stats.heavy_tasks++;
                    bytes[t] = hlists[t].n * (u64)(NW + 1) * 8;                               // ScatteredKmerList::get_size_bytes
                }
            }
            rc = together(cm.allreduce_with_status(bytes, RCCL_SUM, local_rc != 0, c->stream, c->pool), "task sizes");
            if (rc) { parse_release(c, job); return rc; }
            rc = plan_dispatch(bytes.data(), (int)ntasks, nranks, c->cfg.plain_dispatcher != 0, c->cfg.dispatch_upper_coe, c->cfg.dispatch_step, owner.data());
            if (rc) { parse_release(c, job); return fail(c, HSK_ERR_DISPATCH, "%s", hsk_strerror(HSK_ERR_DISPATCH)); }      // (same input on every rank: all of them fail here)
            std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return owner[x] < owner[y]; });
        }
        if (vts) { std::vector<u32> order_v(nvt); for (u32 v = 0; v < nvt; ++v) order_v[v] = v; rc = parse_place(c, job, order_v, st, nullptr, false); }
        else rc = parse_place(c, job, order, st, any_heavy ? &is_heavy : nullptr, nranks > 1);
        parse_release(c, job);
        if (rc && nranks == 1) return rc;
        place_rc = rc;                                           // several ranks: carried into the size-matrix all-reduce below
    }
    pt.end(PH_PARSE);
    c->vt_shift = 0;
    tmark("parse enqueued (task totals read)");
    out->total_supermers = st.tot_sup; out->total_supermer_bytes = st.tot_bytes + st.tot_sup * (ext ? 9 : 1);

    // ---- exchange (multi-GPU) ---------------------------------------------------------------------
    // After this block `segs[t]` lists where the supermers of owned task t live.
    const u8 *x_len = st.sm_len; const u32 *x_pos = st.sm_pos; const int32_t *x_rid = st.sm_rid;
    BaseSource x_src = source_from_store(st, d_packed, packed_bytes);
    ExchangeBuffers xb;
    GroupFeeder feeder; bool fed = false;
    pt.begin(PH_EXCH);
    if (nranks > 1) {
        int local_rc = place_rc ? place_rc : pack_store_bytes(c, st, x_src, !overlap_enabled());
        int rc;
        if (overlap_enabled()) {
            // size matrix: every rank contributes its row, the sum is the full matrix (+ the ranks' status: a rank whose
            // placement or byte packing failed leaves together with its peers)
            std::vector<u64> M((size_t)nranks * ntasks * 3, 0);
            if (!local_rc) for (size_t i = 0; i < (size_t)ntasks * 3; ++i) M[(size_t)rank * ntasks * 3 + i] = st.task_tot[i];
            M.push_back((!local_rc && (st.sm_sub16 != nullptr || st.tot_sup == 0)) ? 1 : 0);      // this rank's supermers carry their minimizer bits (or it has none to send)
            const int st_ = c->comm.allreduce_with_status(M, RCCL_SUM, local_rc != 0, c->stream, c->pool);
            if (st_ < 0) return fail(c, HSK_ERR_COMM, "allreduce(size matrix) failed: %d (%s)", st_, c->comm.last_error.c_str());
            if (st_ > 0) return local_rc ? local_rc : fail(c, HSK_ERR_COMM, "another rank failed before the supermer exchange");
            const bool all_sub = M.back() == (u64)nranks && c->combine_now;
            M.pop_back();
            rc = feeder.plan(c, nranks, rank, ntasks, owner, order, M, st.task_base, segs); if (rc) return rc;
            feeder.st = &st; feeder.lazy_pack = true; feeder.with_sub = all_sub; fed = true;
        } else {
            rc = exchange_supermers(c->comm, c->stream, c->pool, ext, K, ntasks, owner, order, st.task_tot, st.task_base,
                                    st.sm_len, st.sm_bytes, st.sm_pos, st.sm_rid, xb, segs, local_rc != 0);
            if (rc > 0 && local_rc) return local_rc;
            if (rc) return fail(c, HSK_ERR_COMM, "supermer exchange failed: %d (%s)", rc, c->comm.last_error.c_str());
            x_len = xb.len; x_pos = xb.pos; x_rid = xb.rid;
            x_src = source_from_bytes(xb.bytes, xb.nbytes);
            free_store(c, st);
        }
    } else if (!pipelined && vts) {
        for (u32 v = 0; v < nvt; ++v) {
            TaskSegs &ts = segs[v >> vts];
            ts.nkmers += st.task_tot[3 * v + 2];
            if (st.task_tot[3 * v] == 0) continue;
            ExpSeg sg; sg.sup_off = st.task_base[3 * v]; sg.n_sup = st.task_tot[3 * v]; sg.byte_off = v & ((1u << vts) - 1u); sg.kmer_off = 0; sg.tile_start = 0;
            ts.segs.push_back(sg);
        }
    } else if (!pipelined) {
        for (u32 t = 0; t < ntasks; ++t) {
            if (st.task_tot[3 * t] == 0) continue;
            ExpSeg s; s.sup_off = st.task_base[3 * t]; s.n_sup = st.task_tot[3 * t]; s.byte_off = st.task_base[3 * t + 1]; s.kmer_off = 0; s.tile_start = 0;
            segs[t].segs.push_back(s); segs[t].nkmers = st.task_tot[3 * t + 2];
        }
    }
    if (any_heavy) {
        // the k-mer lists of the heavy tasks go to their owners: counts by all-reduce, one grouped send/recv
        std::vector<u32> hv_tasks; for (u32 t = 0; t < ntasks; ++t) if (is_heavy[t]) hv_tasks.push_back(t);
        const size_t nh = hv_tasks.size();
        std::vector<u64> Hn((size_t)nranks * nh, 0);
        for (size_t i = 0; i < nh; ++i) Hn[(size_t)rank * nh + i] = hlists[hv_tasks[i]].n;
        int rc = c->comm.allreduce_sum_u64(Hn.data(), Hn.size(), c->stream, c->pool);
        if (rc) return fail(c, HSK_ERR_COMM, "allreduce(heavy list sizes) failed: %d (%s)", rc, c->comm.last_error.c_str());
        const size_t ew = (size_t)(NW + 1) * 8;
        bool oom = false;
        for (size_t i = 0; i < nh; ++i) {
            const u32 t = hv_tasks[i];
            if (owner[t] != rank) continue;
            HeavyIn hv; hv.task = t; hv.n = 0; hv.d_entries = nullptr;
            for (int p = 0; p < nranks; ++p) hv.n += Hn[(size_t)p * nh + i];
            if (hv.n && !oom) { hv.d_entries = (u64 *)c->pool.alloc(hv.n * ew); if (!hv.d_entries) oom = true; }
            hin.push_back(hv);
        }
        {   // every owner must have its receive buffers before anybody sends
            std::vector<u64> none;
            const int st_ = c->comm.allreduce_with_status(none, RCCL_MAX, oom, c->stream, c->pool);
            if (st_ < 0) return fail(c, HSK_ERR_COMM, "allreduce(status) failed: %d (%s)", st_, c->comm.last_error.c_str());
            if (st_ > 0) { for (auto &hv : hin) c->pool.release(hv.d_entries); hin.clear(); return oom ? fail(c, HSK_ERR_OOM, "heavy-hitter receive buffers") : fail(c, HSK_ERR_COMM, "another rank ran out of memory before the heavy-hitter exchange"); }
        }
        Comm &cm = c->comm;
        size_t hi = 0;
        {
            P2PGroup grp(cm);
            for (size_t i = 0; i < nh && !grp.rc; ++i) {
                const u32 t = hv_tasks[i];
                if (owner[t] == rank) {
                    HeavyIn &hv = hin[hi++];
                    u64 o = 0;
                    for (int p = 0; p < nranks; ++p) {
                        const u64 n = Hn[(size_t)p * nh + i];
                        if (n && p != rank) grp.recv((char *)hv.d_entries + o * ew, n * ew, p, c->stream, "ncclRecv(heavy list)");
                        o += n;
                    }
                } else if (hlists[t].n) {
                    grp.send(hlists[t].entries, hlists[t].n * ew, owner[t], c->stream, "ncclSend(heavy list)");
                }
            }
            if (grp.end()) return fail(c, HSK_ERR_COMM, "heavy-hitter list exchange failed: %s", cm.last_error.c_str());
        }
        hi = 0;
        for (size_t i = 0; i < nh; ++i) {                                   // own share: device copy
            const u32 t = hv_tasks[i];
            if (owner[t] != rank) continue;
            HeavyIn &hv = hin[hi++];
            u64 o = 0; for (int p = 0; p < rank; ++p) o += Hn[(size_t)p * nh + i];
            if (hlists[t].n) HIPCHK(c, hipMemcpyAsync((char *)hv.d_entries + o * ew, hlists[t].entries, hlists[t].n * ew, hipMemcpyDeviceToDevice, c->stream));
        }
        HIPCHK(c, hsk_sync(c, c->stream));
        for (auto &to : hlists) free_task_out(c, to);
    }
    pt.end(PH_EXCH);
    ProcExtra ex; ex.heavy_in = &hin; ex.vt_shift = vts; if (nranks == 1 && st.sm_item) ex.items_store = &st;
    const std::vector<void *> before_rank = (fed && c->comm.active()) ? c->pool.snapshot() : std::vector<void *>();
    int rc = process_rank<NW>(c, ntasks, owner, rank, segs, x_len, x_src, x_pos, x_rid, out, rp, pt, true, fed ? &feeder : nullptr, &ex);
    if (rc == HSK_OK && c->dropped_now) { out->total_kmers += c->dropped_now; c->stats.dropped_kmers += (int64_t)c->dropped_now; }      // (the instances the scan left out are k-mers of the input all the same)
    if (rc == HSK_OK && nranks > 1) for (u32 t = 0; t < ntasks; ++t) if (is_heavy[t] && owner[t] == rank) out->total_kmers += heavy_kmers[t];      // (they arrived as lists, not as supermers)
    if (fed && feeder.live) {
        // Leaving together, part two (part one: the all-reduces with status up to the first task group).  A rank whose count failed
        // while the groups were travelling has kept its side of the exchange going (drain_after_failure); now the ranks tell each
        // other how it went, and if one of them failed they all return an error -- the failed rank its own, the others HSK_ERR_COMM.
        if (rc != HSK_OK) {
            char keep_msg[sizeof c->err]; memcpy(keep_msg, c->err, sizeof keep_msg);
            const int drc = feeder.drain_after_failure(before_rank);
            if (drc) return drc;                                                   // the transport itself is broken: nothing more to agree on
            memcpy(c->err, keep_msg, sizeof keep_msg);
        }
        std::vector<u64> none;
        const int st_ = c->comm.allreduce_with_status(none, RCCL_MAX, rc != HSK_OK, c->stream, c->pool);
        if (st_ < 0) return fail(c, HSK_ERR_COMM, "allreduce(final status) failed: %d (%s)", st_, c->comm.last_error.c_str());
        if (st_ > 0 && rc == HSK_OK) rc = fail(c, HSK_ERR_COMM, "another rank failed while the supermers were travelling; this rank's result is dropped");
    }
    for (auto &hv : hin) c->pool.release(hv.d_entries);
    if (nranks > 1 && !fed) xb.release(c->pool); else free_store(c, st);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// virtual ranks on one GPU: the multi-GPU data path (probe, dispatch, owner-grouped parse, pack,
// all-to-all-v plan, multi-segment expand) with device-to-device copies in place of RCCL send/recv.
// This is how the exchange logic is exercised on a single-GPU box (tests/test_gpu_multirank.py).
// ------------------------------------------------------------------------------------------------
struct DevInput { u8 *packed = nullptr; u64 *roff = nullptr; u32 *rlen = nullptr; };

template <int NW>
static int run_loopback(hsk_ctx *c, int R, const DevInput *in, const u64 *packed_bytes, const u64 *nreads, hsk_result *outs, int32_t *owner_out, u32 *ntasks_out)
{
    const bool ext = c->cfg.extension != 0;
    u64 tot_bytes = 0; for (int r = 0; r < R; ++r) tot_bytes += packed_bytes[r];
    const u32 ntasks = c->cfg.ntasks ? (u32)c->cfg.ntasks : auto_ntasks(c, tot_bytes / (u64)R + 1, R);
    *ntasks_out = ntasks;
    std::vector<int64_t> rid_base(R, 0);
    for (int r = 1; r < R; ++r) rid_base[r] = rid_base[r - 1] + (int64_t)nreads[r - 1];      // MPI_Exscan of the read counts
    std::vector<u32> order(ntasks); for (u32 t = 0; t < ntasks; ++t) order[t] = t;
    // the plan, as run_pipeline chooses it with several ranks: the sketch of a rank's reads (here: of the first virtual rank that has some)
    c->combine_now = false; c->item_mode_now = false; c->vt_shift = 0; c->combine_left_now = false; c->drop_mask_now = 0; c->dropped_now = 0;
    const bool scan_ok = parse_fast_enabled() && c->cfg.minimizer_size <= SCAN_MAX_M;
    const bool plan_cond = (NW == 1 || (NW == 2 && c->cfg.kmer_size >= 40 && c->cfg.kmer_size <= 55)) && R > 1 && !ext && combine_enabled() && scan_ok && c->xcd_batch_ok && overlap_enabled() && place_bytes_enabled(true);
    if (R > 1 && (plan_cond || (scan_ok && c->cfg.kmer_size <= 57))) {          // (without a plan to choose, the sketch still says which k-mers are certain to be dropped)
        const u64 combine_min = (u64)tune("combine_min_bytes", 64LL << 20);
        int r0 = 0; while (r0 + 1 < R && nreads[r0] == 0) ++r0;
        int erc = estimate_plan(c, in[r0].packed, packed_bytes[r0], in[r0].roff, in[r0].rlen, nreads[r0], R); if (erc) return erc;
        c->drop_mask_now = certain_drop_mask(c);              // (a rank that is certain is right for all: the virtual ranks share the first one's)
        if (plan_cond) {
            const bool pays = c->est.valid ? c->est.distinct_per_kmer * c->est_bias * (double)combine_ratio() <= 1.0 : !c->combine_off;
            c->combine_now = pays && tot_bytes / (u64)R >= combine_min && !(NW == 1 ? c->agg_off : c->agg_off_wide);
        }
    }
    struct PlanReset { hsk_ctx *c; ~PlanReset() { c->combine_now = false; c->est.valid = false; c->drop_mask_now = 0; } } plan_reset{c};
    // 1. hash every rank's reads once (parse_count), sum the task sizes, dispatch
    std::vector<u64> bytes(ntasks, 0), dropped(R, 0);
    std::vector<ParseJob> jobs(R);
    auto release_jobs = [&]() { for (auto &j : jobs) parse_release(c, j); };
    // per-rank device time of the parse (hash + count, then placement + byte store): outs[r].ms_parse
    EvList tev(c);
    std::vector<hipEvent_t> e0(R), e1(R), e2(R), e3(R);
    for (int r = 0; r < R; ++r) { e0[r] = tev.get(); e1[r] = tev.get(); e2[r] = tev.get(); e3[r] = tev.get(); }
    auto parse_ms = [&](int r) { float a = 0, b = 0; (void)hipEventElapsedTime(&a, e0[r], e1[r]); (void)hipEventElapsedTime(&b, e2[r], e3[r]); return (double)a + (double)b; };
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e0[r], c->stream);
        int rc = parse_count(c, in[r].packed, packed_bytes[r], in[r].roff, in[r].rlen, nreads[r], rid_base[r], ntasks, jobs[r]);
        dropped[r] = c->dropped_now;
        (void)hipEventRecord(e1[r], c->stream);
        if (rc) { release_jobs(); return rc; }
        for (u32 t = 0; t < ntasks; ++t) bytes[t] += jobs[r].task_tot[3 * t + 1] + jobs[r].task_tot[3 * t] * (ext ? 9 : 1);
    }
    // 1b. heavy-hitter tasks: classify on the global k-mer counts, every rank pre-aggregates its share
    std::vector<u8> is_heavy(ntasks, 0);
    std::vector<u64> heavy_kmers(ntasks, 0);             // k-mer instances of a heavy task over all ranks: its owner counts them into total_kmers
    std::vector<std::vector<TaskOut>> hlists(R);
    auto free_hlists = [&]() { for (auto &v : hlists) for (auto &to : v) free_task_out(c, to); };
    bool any_heavy = false;
    if (heavy_enabled(c, NW, R)) {
        std::vector<u64> kg(ntasks, 0); std::vector<int32_t> types(ntasks, 0);
        for (int r = 0; r < R; ++r) for (u32 t = 0; t < ntasks; ++t) kg[t] += jobs[r].task_tot[3 * t + 2];
        plan_classify(kg.data(), (int)ntasks, c->cfg.unbalanced_ratio, types.data());
        for (u32 t = 0; t < ntasks; ++t) if (types[t] == 1) { is_heavy[t] = 1; any_heavy = true; heavy_kmers[t] = kg[t]; }
    }
    if (any_heavy) {
        std::vector<u8> bad(ntasks, 0);
        for (int r = 0; r < R; ++r) {
            std::vector<u8> failed;
            int rc = heavy_preaggregate<NW>(c, jobs[r], in[r].packed, packed_bytes[r], is_heavy, hlists[r], failed);
            if (rc) { release_jobs(); free_hlists(); return rc; }
            for (u32 t = 0; t < ntasks; ++t) bad[t] |= failed[t];
        }
        any_heavy = false;
        for (u32 t = 0; t < ntasks; ++t) {
            if (!is_heavy[t]) continue;
            if (bad[t]) { is_heavy[t] = 0; for (int r = 0; r < R; ++r) free_task_out(c, hlists[r][t]); continue; }   // travels as supermers after all
            any_heavy = true; c->stats.heavy_tasks++;
            bytes[t] = 0;
            for (int r = 0; r < R; ++r) bytes[t] += hlists[r][t].n * (u64)(NW + 1) * 8;       // ScatteredKmerList::get_size_bytes
        }
    }
    std::vector<int32_t> owner(ntasks, 0);
    if (plan_dispatch(bytes.data(), (int)ntasks, R, c->cfg.plain_dispatcher != 0, c->cfg.dispatch_upper_coe, c->cfg.dispatch_step, owner.data())) {
        release_jobs(); free_hlists();
        return fail(c, HSK_ERR_DISPATCH, "%s", hsk_strerror(HSK_ERR_DISPATCH));
    }
    if (owner_out) memcpy(owner_out, owner.data(), sizeof(int32_t) * ntasks);
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return owner[x] < owner[y]; });
    // 2. owner-grouped placement + byte materialisation on every rank
    std::vector<SupermerStore> st(R);
    std::vector<u64> M((size_t)R * ntasks * 3, 0);
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e2[r], c->stream);
        int rc = parse_place(c, jobs[r], order, st[r], any_heavy ? &is_heavy : nullptr, R > 1);
        parse_release(c, jobs[r]);
        if (rc) { release_jobs(); return rc; }
        rc = pack_store_bytes(c, st[r], source_from_packed(in[r].packed, packed_bytes[r], st[r].sm_gpos), !overlap_enabled()); if (rc) { release_jobs(); return rc; }
        (void)hipEventRecord(e3[r], c->stream);
        for (size_t i = 0; i < (size_t)ntasks * 3; ++i) M[(size_t)r * ntasks * 3 + i] = st[r].task_tot[i];
    }
    // 2b. the k-mer lists of the heavy tasks go to their owners (device copies here, send/recv in run_pipeline)
    std::vector<std::vector<HeavyIn>> hin(R);
    if (any_heavy) {
        for (u32 t = 0; t < ntasks; ++t) {
            if (!is_heavy[t]) continue;
            HeavyIn hv; hv.task = t; hv.n = 0; hv.d_entries = nullptr;
            for (int r = 0; r < R; ++r) hv.n += hlists[r][t].n;
            if (hv.n) {
                DALLOC(c, hv.d_entries, u64 *, hv.n * (NW + 1) * 8);
                u64 o = 0;
                for (int r = 0; r < R; ++r) {
                    if (hlists[r][t].n) HIPCHK(c, hipMemcpyAsync(hv.d_entries + o * (NW + 1), hlists[r][t].entries, hlists[r][t].n * (NW + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
                    o += hlists[r][t].n;
                }
            }
            hin[owner[t]].push_back(hv);
        }
        HIPCHK(c, hsk_sync(c, c->stream));
        free_hlists();
    }
    auto free_hin = [&]() { for (auto &v : hin) for (auto &hv : v) c->pool.release(hv.d_entries); };
    // 3. the exchange: same plans as the RCCL path (hsk_comm.h), device copies instead of send/recv
    if (overlap_enabled()) {
        // grouped exchange overlapped with the sort, exactly as run_pipeline drives it
        std::vector<GroupFeeder> fd(R);
        std::vector<std::vector<ExchangePlan>> pl_all(R);
        std::vector<std::vector<TaskSegs>> segs(R);
        bool all_sub = c->combine_now;
        for (int r = 0; r < R; ++r) if (st[r].tot_sup && !st[r].sm_sub16) all_sub = false;
        for (int d = 0; d < R; ++d) { int rc = fd[d].plan(c, R, d, ntasks, owner, order, M, st[d].task_base, segs[d]); if (rc) return rc; pl_all[d] = fd[d].pl; fd[d].with_sub = all_sub; }
        int rc_all = HSK_OK;
        for (int r = 0; r < R && rc_all == HSK_OK; ++r) {
            fd[r].st_all = &st; fd[r].pl_all = &pl_all; fd[r].lazy_pack = true;
            memset(&outs[r], 0, sizeof(hsk_result));
            ResultPriv *rp = new ResultPriv();
            outs[r].priv = rp; outs[r].nw = NW; outs[r].ntasks = (int32_t)ntasks;
            PhaseTimer pt(c);
            ProcExtra ex; ex.heavy_in = &hin[r];
            rc_all = process_rank<NW>(c, ntasks, owner, r, segs[r], nullptr, BaseSource(), nullptr, nullptr, &outs[r], rp, pt, false, &fd[r], &ex);
            if (rc_all == HSK_OK) { outs[r].total_kmers += dropped[r]; c->stats.dropped_kmers += (int64_t)dropped[r]; }
            if (rc_all == HSK_OK) for (u32 t = 0; t < ntasks; ++t) if (is_heavy[t] && owner[t] == r) outs[r].total_kmers += heavy_kmers[t];      // (they arrived as lists, not as supermers)
            if (rc_all == HSK_OK) { outs[r].ms_parse = parse_ms(r); outs[r].ms_total = outs[r].ms_parse + outs[r].ms_exchange + outs[r].ms_extract + outs[r].ms_sort + outs[r].ms_count + outs[r].ms_d2h; }
        }
        for (int r = 0; r < R; ++r) free_store(c, st[r]);
        free_hin();
        return rc_all;
    }
    std::vector<ExchangePlan> pl(R);
    std::vector<std::vector<TaskSegs>> segs(R);
    std::vector<ExchangeBuffers> xb(R);
    for (int d = 0; d < R; ++d) {
        plan_exchange(R, d, ntasks, owner, order, M, st[d].task_base, pl[d], segs[d]);
        xb[d].len = (u8 *)c->pool.alloc(pl[d].recv_tot_sup + 64); xb[d].bytes = (u8 *)c->pool.alloc(pl[d].recv_tot_bytes + 64); xb[d].nbytes = pl[d].recv_tot_bytes;
        if (ext) { xb[d].pos = (u32 *)c->pool.alloc(pl[d].recv_tot_sup * 4 + 64); xb[d].rid = (int32_t *)c->pool.alloc(pl[d].recv_tot_sup * 4 + 64); }
        if (!xb[d].len || !xb[d].bytes || (ext && (!xb[d].pos || !xb[d].rid))) return fail(c, HSK_ERR_OOM, "exchange buffers");
    }
    for (int d = 0; d < R; ++d) for (int sidx = 0; sidx < R; ++sidx) {
        const u64 n = pl[sidx].send_sup[d], nb = pl[sidx].send_bytes[d];
        if (n != pl[d].recv_sup[sidx] || nb != pl[d].recv_bytes[sidx]) return fail(c, HSK_ERR_INTERNAL, "exchange plan mismatch %d->%d", sidx, d);
        if (!n) continue;
        HIPCHK(c, hipMemcpyAsync(xb[d].len + pl[d].recv_sup_off[sidx], st[sidx].sm_len + pl[sidx].send_sup_off[d], n, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(xb[d].bytes + pl[d].recv_byte_off[sidx], st[sidx].sm_bytes + pl[sidx].send_byte_off[d], nb, hipMemcpyDeviceToDevice, c->stream));
        if (ext) {
            HIPCHK(c, hipMemcpyAsync(xb[d].pos + pl[d].recv_sup_off[sidx], st[sidx].sm_pos + pl[sidx].send_sup_off[d], n * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(xb[d].rid + pl[d].recv_sup_off[sidx], st[sidx].sm_rid + pl[sidx].send_sup_off[d], n * 4, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    HIPCHK(c, hsk_sync(c, c->stream));
    for (int r = 0; r < R; ++r) free_store(c, st[r]);
    // 4. every rank finishes its own tasks
    for (int r = 0; r < R; ++r) {
        memset(&outs[r], 0, sizeof(hsk_result));
        ResultPriv *rp = new ResultPriv();
        outs[r].priv = rp; outs[r].nw = NW; outs[r].ntasks = (int32_t)ntasks;
        PhaseTimer pt(c);
        ProcExtra ex; ex.heavy_in = &hin[r];
        int rc = process_rank<NW>(c, ntasks, owner, r, segs[r], xb[r].len, source_from_bytes(xb[r].bytes, xb[r].nbytes), xb[r].pos, xb[r].rid, &outs[r], rp, pt, false, nullptr, &ex);
        xb[r].release(c->pool);
        if (rc) { free_hin(); return rc; }
        outs[r].ms_parse = parse_ms(r); outs[r].ms_total = outs[r].ms_parse + outs[r].ms_exchange + outs[r].ms_extract + outs[r].ms_sort + outs[r].ms_count + outs[r].ms_d2h;
    }
    free_hin();
    return HSK_OK;
}
