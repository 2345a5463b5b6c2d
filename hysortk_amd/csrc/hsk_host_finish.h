// hsk_host_finish.h -- host side of the count stage: two-pass counter, fused tile finish, aggregating finishes (kernels: hsk_count.h, hsk_finish.h, hsk_agg.h).
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// ------------------------------------------------------------------------------------------------
// stage: merge-count one sorted task (a13)
// ------------------------------------------------------------------------------------------------
struct TaskOut { u64 n = 0, npay = 0; u64 *entries = nullptr; u64 *payoff = nullptr; u32 *pos = nullptr; int32_t *rid = nullptr; bool failed = false; u64 pay_base = 0; };

template <int NW>
static int count_task_device(hsk_ctx *c, const u64 *keys, const u64 *vals, u64 n, u64 payoff_add, u64 *d_histo, u32 histo_len, TaskOut &out)
{
    out = TaskOut();
    if (n == 0) return HSK_OK;
    const bool ext = vals != nullptr;
    const u64 ntiles = (n + CNT_TILE - 1) / CNT_TILE;
    u64 *d_tile_cnt, *d_total;
    DALLOC(c, d_tile_cnt, u64 *, ntiles * 8);
    DALLOC(c, d_total, u64 *, 256);
    CountArgs a; memset(&a, 0, sizeof a);
    a.keys = keys; a.n = n; a.lower = (u32)c->cfg.lower_freq; a.upper = (u32)c->cfg.upper_freq;
    a.tile_cnt = d_tile_cnt; a.histo = d_histo; a.histo_len = histo_len; a.payoff_add = payoff_add;
    hipLaunchKernelGGL((count_kernel<NW, false, false>), dim3((u32)ntiles), dim3(CNT_THREADS), 0, c->stream, a);
    hipLaunchKernelGGL(count_scan_kernel, dim3(1), dim3(CNT_THREADS), 0, c->stream, d_tile_cnt, ntiles, d_total);
    u64 *tot = (u64 *)((char *)c->pinned + c->pinned_bytes - 128);
    HIPCHK(c, hipMemcpyAsync(tot, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));
    out.n = tot[0]; out.npay = ext ? n : 0;
    if (ext) {
        // the payload of a kept run is its slice of the sorted payload array: split the whole array once
        DALLOC(c, out.pos, u32 *, n * 4);
        DALLOC(c, out.rid, int32_t *, n * 4);
        hipLaunchKernelGGL(payload_split_kernel, dim3((u32)std::min<u64>((n + 255) / 256, 4096)), dim3(256), 0, c->stream, vals, n, out.pos, out.rid);
    }
    if (out.n) {
        DALLOC(c, out.entries, u64 *, out.n * (NW + 1) * 8);
        if (ext) DALLOC(c, out.payoff, u64 *, out.n * 8);
        a.entries = out.entries; a.run_start = out.payoff;
        // persistent: one histogram flush per workgroup; exactly the resident workgroup count, so no ragged second wave
        static int occ_e[2] = {0, 0};
        int &occ = occ_e[ext ? 1 : 0];
        if (!occ) {
            int nb = 0;
            hipError_t e = ext ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, count_kernel<NW, true, true>, CNT_THREADS, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, count_kernel<NW, true, false>, CNT_THREADS, 0);
            occ = (e == hipSuccess && nb > 0) ? nb : 4;
        }
        hipDeviceProp_t *pr = nullptr; (void)pr;
        const u32 egrid = (u32)std::min<u64>(ntiles, (u64)occ * 256);
        if (ext) hipLaunchKernelGGL((count_kernel<NW, true, true>), dim3(egrid), dim3(CNT_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((count_kernel<NW, true, false>), dim3(egrid), dim3(CNT_THREADS), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    c->pool.release(d_tile_cnt); c->pool.release(d_total);
    return HSK_OK;
}

static void free_task_out(hsk_ctx *c, TaskOut &o)
{
    c->pool.release(o.entries); c->pool.release(o.payoff); c->pool.release(o.pos); c->pool.release(o.rid);
    o = TaskOut();
}

// ------------------------------------------------------------------------------------------------
// the whole path
// ------------------------------------------------------------------------------------------------
struct ResultPriv {
    std::vector<void *> host_blocks;     // hipHostMalloc'ed
    std::vector<TaskOut> dev_tasks;      // kept in HBM with HSK_FLAG_KEEP_DEVICE
};

static void *host_alloc(hsk_ctx *c, ResultPriv *rp, size_t bytes)
{
    void *p = c->hpool.alloc(bytes);
    if (p) rp->host_blocks.push_back(p);
    return p;
}
static void host_release(hsk_ctx *c, ResultPriv *rp, void *p)
{
    for (size_t i = 0; i < rp->host_blocks.size(); ++i) if (rp->host_blocks[i] == p) { rp->host_blocks.erase(rp->host_blocks.begin() + i); break; }
    c->hpool.release(p);
}

static bool finish_enabled();
static bool agg_enabled();
static u32 auto_ntasks(hsk_ctx *c, u64 packed_bytes, int nranks)
{
    // one task per ~2^28 k-mers (2 GB of 8-byte keys): large enough to saturate the chip, small
    // enough that key + ping-pong + look-back buffers of one task stay a small share of HBM
    u64 est = packed_bytes * 4 * (u64)std::max(nranks, 1);
    u64 t = (est + (1ULL << 28) - 1) >> 28;
    t = std::max<u64>(t, (u64)std::max(nranks, 1));
    // tasks are sorted eight at a time (one per XCD): give every rank a multiple of eight when there are that many -- and eight
    // as soon as the rank has ~2^25 base positions: from there on the batch path (expand fused with the first pass, aggregation)
    // beats one task on the single-task path (measured: 40 M k-mers 2.1 vs 2.4 ms, 400 M 7.8 vs 18.2 ms; 4 M 1.0 vs 0.7 ms)
    const u64 per = 8ULL * (u64)std::max(nranks, 1);
    if (packed_bytes * 4 >= (1ULL << 25)) t = std::max(t, per);
    if (t >= per) t = (t + per - 1) / per * per;
    return (u32)std::min<u64>(std::max<u64>(t, 1), HSK_MAX_TASKS);
}

// Fused finish of a batch (hybrid sort, one-word keys, no payload): one finish_multi_kernel launch turns the
// prefix-ordered keys of eight tasks into their (k-mer, count) lists.  Tasks the kernel could not finish (a
// long bin with several keys, see hsk_finish.h) are redone with the full-width passes and the two-pass counter.
static bool finish_enabled()
{
    return !(g_plan_flags & HSK_FLAG_FULL_SORT);
}

template <int NW>
static int finish_batch_device(hsk_ctx *c, BatchTask *bt, int K, u64 max_task, u64 *d_histo, u32 histo_len, TaskOut *outs)
{
    static_assert(NW == 1, "fused finish handles one-word keys");
    const u32 L = (u32)c->cfg.lower_freq;
    const u32 cap_t = (u32)FN_NL / L + 1;                  // a tile keeps at most (2048 + 512) / L runs
    u64 ntiles[XCD_BATCH], cnt_off[XCD_BATCH + 1]; cnt_off[0] = 0;
    for (int i = 0; i < XCD_BATCH; ++i) { ntiles[i] = (bt[i].n + FN_TILE - 1) / FN_TILE; cnt_off[i + 1] = cnt_off[i] + ntiles[i] + 1; }
    // control block: [8] flags (u32), then per task the tile counts (+1 word for the total)
    const size_t ctl_bytes = 64;
    char *d_ctl = (char *)c->pool.alloc(ctl_bytes + cnt_off[XCD_BATCH] * 8 + 64);
    if (!d_ctl) return fail(c, HSK_ERR_OOM, "finish control block");
    HIPCHK(c, hipMemsetAsync(d_ctl, 0, ctl_bytes, c->stream));
    u32 *d_flags = (u32 *)d_ctl; u64 *d_cnt = (u64 *)(d_ctl + ctl_bytes);
    // scratch: the idle ping-pong buffer of the task when the per-tile slots fit into it (L >= 3), else its own block
    u64 *scratch[XCD_BATCH] = {nullptr}; bool own_scratch[XCD_BATCH] = {false};
    for (int i = 0; i < XCD_BATCH; ++i) {
        if (bt[i].n == 0) continue;
        u64 *other = (bt[i].out_k == bt[i].kA) ? bt[i].kB : bt[i].kA;
        const u64 need = ntiles[i] * (u64)cap_t * 16;
        if (need <= max_task * 8) scratch[i] = other;
        else { scratch[i] = (u64 *)c->pool.alloc(need + 64); own_scratch[i] = true; if (!scratch[i]) return fail(c, HSK_ERR_OOM, "finish scratch of %llu bytes", (unsigned long long)need); }
        FinishArgs a; memset(&a, 0, sizeof a);
        a.keys = bt[i].out_k; a.n = bt[i].n; a.scratch = scratch[i]; a.cap_t = cap_t; a.tile_cnt = d_cnt + cnt_off[i]; a.flags = d_flags + i;
        a.lower = L; a.upper = (u32)c->cfg.upper_freq; a.hi_shift = HYBRID_SHIFT;
        hipLaunchKernelGGL(finish_kernel, dim3((u32)ntiles[i]), dim3(FN_THREADS), 0, c->stream, a);
        hipLaunchKernelGGL(count_scan_kernel, dim3(1), dim3(CNT_THREADS), 0, c->stream, d_cnt + cnt_off[i], ntiles[i], d_cnt + cnt_off[i] + ntiles[i]);
    }
    HIPCHK(c, hipGetLastError());
    struct { u32 flags[8]; u64 total[8]; } h; memset(&h, 0, sizeof h);
    HIPCHK(c, hipMemcpyAsync(h.flags, d_flags, sizeof h.flags, hipMemcpyDeviceToHost, c->stream));
    for (int i = 0; i < XCD_BATCH; ++i) if (bt[i].n) HIPCHK(c, hipMemcpyAsync(&h.total[i], d_cnt + cnt_off[i] + ntiles[i], 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));
    static int occ = 0;
    if (!occ) { int nb = 0; occ = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, finish_compact_kernel, FN_THREADS, 0) == hipSuccess && nb > 0) ? nb : 4; }
    int rc = HSK_OK;
    for (int i = 0; i < XCD_BATCH && rc == HSK_OK; ++i) {
        outs[i] = TaskOut();
        if (bt[i].n == 0) continue;
        if (h.flags[i] && c->forbid_long_way) { outs[i].failed = true; continue; }
        if (h.flags[i]) {
            // the long way for this task: full-width passes from the current order, then the two-pass counter
            c->stats.redone_tasks++;
            if (own_scratch[i]) { c->pool.release(scratch[i]); scratch[i] = nullptr; own_scratch[i] = false; }
            SortScratch sc1; rc = alloc_sort_scratch(c, sc1); if (rc) break;
            u64 *cur = bt[i].out_k, *other = (cur == bt[i].kA) ? bt[i].kB : bt[i].kA, *sk, *sv;
            rc = sort_task_device<NW>(c, cur, other, nullptr, nullptr, bt[i].n, K, sc1, &sk, &sv, false);
            free_sort_scratch(c, sc1);
            if (rc == HSK_OK) rc = count_task_device<NW>(c, sk, nullptr, bt[i].n, 0, d_histo, histo_len, outs[i]);
            continue;
        }
        c->stats.fused_tasks++;
        outs[i].n = h.total[i];
        if (outs[i].n) {
            outs[i].entries = (u64 *)c->pool.alloc(outs[i].n * 16);
            if (!outs[i].entries) { rc = fail(c, HSK_ERR_OOM, "task output of %llu bytes", (unsigned long long)(outs[i].n * 16)); break; }
            const u32 grid = (u32)std::min<u64>((ntiles[i] + 3) / 4, (u64)occ * 256);
            hipLaunchKernelGGL(finish_compact_kernel, dim3(grid), dim3(FN_THREADS), 0, c->stream, scratch[i], cap_t, d_cnt + cnt_off[i], ntiles[i],
                               outs[i].entries, d_histo, histo_len);
        }
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hsk_sync(c, c->stream));         // scratch buffers are reused by the next batch
    for (int i = 0; i < XCD_BATCH; ++i) if (own_scratch[i]) c->pool.release(scratch[i]);
    c->pool.release(d_ctl);
    return rc;
}

// ---- two passes + aggregation (hsk_agg.h): the batch's keys are sorted on their top 16 bits ---------------
static bool agg_enabled()
{
    return !(g_plan_flags & (HSK_FLAG_NO_AGGREGATION | HSK_FLAG_FULL_SORT));
}

// The aggregating finish of a batch in two stages, so that the host never has to wait for the GPU with nothing queued
// behind the wait:
//   agg_stage1  launches bin bounds, the aggregation with the first-choice table and the bin-count scan, asks for the
//               per-task flags and totals (pinned slot) and records an event;
//   agg_stage2  (called after the NEXT batch's expand / scatter / stage 1 have been enqueued) waits for that event,
//               retries the tasks whose bins overflowed with the large table, sizes the outputs exactly, launches the
//               compaction and sends tasks the tables cannot take the long way.
// prefix_bits = 16: bins of the top 16 bits (two scatter passes), small tables with a retry ladder and the long way.
struct AggHostRead { u32 flags[AG_BATCH]; u32 maxd[AG_BATCH]; u32 ovf[2][AG_BATCH]; u64 total[AG_BATCH]; };   // mirrors the device control block (+ totals)
struct AggPending {
    bool active = false;
    BatchTask bt[AG_BATCH];
    AggArgs a;
    u64 *d_bounds = nullptr, *d_cnt = nullptr, *d_off = nullptr; u32 *d_flags = nullptr;     // d_flags: {flags[8], maxd[8], overflow-list lengths [2][8]}
    u32 *d_list[2] = {nullptr, nullptr};                                    // [AG_BATCH][nbins] overflowing bins, ping-pong between the rungs
    char *d_large = nullptr;                                                // bins of very many records (hsk_agg.h: AggLarge): the tasks' structs, tables and slice lists
    bool own_scratch[AG_BATCH] = {false};
    bool big = false; int first_cap = AG_LOG2CAP_SMALL;
    bool weighted = false;                              // the records are {key, count} pairs (combining extraction): counts are added, no long way
    u32 nbins = 0, slot_shift = 0; int K = 0; u64 ntot = 0;
    hipEvent_t ev = nullptr;
    AggHostRead *h = nullptr;                           // pinned
};
// The long way for SEVERAL tasks of a batch at once (input with nearly unique k-mers sends whole batches there): the
// full-width passes as eight-task launches from the tasks' current order (one task at a time they run at well under half the
// rate), then the two-pass counter per task.  redo[i]: task i takes it; pay_before: EXTENSION (null otherwise).
template <int NW>
static int long_way_batch(hsk_ctx *c, const BatchTask *bt, const bool *redo, int K, const u64 *pay_before, u64 *d_histo, u32 histo_len, TaskOut *outs)
{
    BatchTask b2[XCD_BATCH];
    for (int i = 0; i < XCD_BATCH; ++i) {
        b2[i] = bt[i];
        if (!redo[i]) { b2[i].n = 0; continue; }
        if (bt[i].out_k != bt[i].kA) { std::swap(b2[i].kA, b2[i].kB); std::swap(b2[i].vA, b2[i].vB); }      // the current order becomes the A side
        c->stats.redone_tasks++;
    }
    const auto counted = c->stats.redone_tasks;
    int rc = sort_batch_device<NW>(c, b2, K, false);
    c->stats.redone_tasks = counted;                        // (a task the prefix plan of these passes has to redo once more is still one redone task)
    for (int i = 0; i < XCD_BATCH && rc == HSK_OK; ++i)
        if (redo[i]) rc = count_task_device<NW>(c, b2[i].out_k, b2[i].out_v, bt[i].n, pay_before ? pay_before[i] : 0, d_histo, histo_len, outs[i]);
    return rc;
}

constexpr int AG_LOG2CAP_HUGE = 13;                     // last rung for one-word keys: agg_big_kernel (8192 slots, 1024 threads) on the listed bins

// One rung of the ladder: the aggregation kernel with a 2^log2cap table over all bins (grid_x = nbins) or over the listed
// bins (grid_x = the longest list).
template <int NW>
static int agg_launch_rung(hsk_ctx *c, AggPending &p, int log2cap, u32 grid_x, u64 records)
{
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    const AggArgs &a = p.a;
    EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 2; ep.keys = records; ep.bytes = records * (NW * 8 + (p.weighted ? 8 : 0)); (void)hipEventRecord(ep.a, c->stream); }      // (record bytes READ: keys, and the counts of {k-mer, count} pairs)
    if (NW == 3) {
        if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg3_finish_kernel<AG_LOG2CAP_SMALL>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else if (log2cap == AG_LOG2CAP_MEDIUM) hipLaunchKernelGGL((agg3_finish_kernel<AG_LOG2CAP_MEDIUM>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((agg3_finish_kernel<AG_LOG2CAP_LARGE>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    } else if (NW == 2) {
        if (p.weighted) {
            if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_SMALL, true>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
            else if (log2cap == AG_LOG2CAP_MEDIUM) hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_MEDIUM, true>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
            else hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_LARGE, true>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        }
        else if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_SMALL>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else if (log2cap == AG_LOG2CAP_MEDIUM) hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_MEDIUM>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_LARGE>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    } else
    if (p.weighted) {
        if constexpr (NW == 1) {
            if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_SMALL, true>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
            else if (log2cap == AG_LOG2CAP_MEDIUM) hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_MEDIUM, true>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
            else hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_LARGE, true>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        }
    } else
    if (p.big || log2cap == AG_LOG2CAP_HUGE) hipLaunchKernelGGL(agg_big_kernel, dim3(grid_x, AG_BATCH), dim3(AGB_THREADS), 0, c->stream, a);
    else if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_SMALL>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    else if (log2cap == AG_LOG2CAP_MEDIUM) hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_MEDIUM>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    else hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_LARGE>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
    HIPCHK(c, hipGetLastError());
    return HSK_OK;
}
// bin-count scan + read-back of the control block and the totals (asynchronous: the caller waits)
static int agg_launch_scan(hsk_ctx *c, AggPending &p)
{
    hipLaunchKernelGGL(agg_scan_kernel, dim3(AG_BATCH), dim3(AG_THREADS), 0, c->stream, p.a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(p.h->flags, p.d_flags, 4 * sizeof(u32) * AG_BATCH, hipMemcpyDeviceToHost, c->stream));
    // the eight totals sit behind the last bin of every task's count row: one strided copy
    HIPCHK(c, hipMemcpy2DAsync(p.h->total, 8, p.d_off + p.nbins, ((size_t)p.nbins + 8) * 8, 8, AG_BATCH, hipMemcpyDeviceToHost, c->stream));
    return HSK_OK;
}

// slot: which of the two pinned read-back areas (two batches can be between their stages at once)
template <int NW>
static int agg_stage1(hsk_ctx *c, const BatchTask *bt, int K, int prefix_bits, int slot, AggPending &p, bool weighted = false)
{
    static_assert(NW <= 3, "the aggregating finish handles keys of one to three words");
    constexpr u32 EW = NW + 1;                          // words per entry
    p = AggPending();
    const u32 L = (u32)c->cfg.lower_freq;
    p.weighted = weighted;
    p.slot_shift = (L >= 2 && !weighted) ? 1 : 0;       // a bin of n records keeps at most n / L entries (pairs: every record may be an entry)
    p.big = false;                                      // (8-bit bins: the one-pass experiment of rounds 1-2, removed)
    p.nbins = 1u << prefix_bits; p.K = K;
    p.h = (AggHostRead *)((char *)c->pinned + c->pinned_bytes - 4096 + (size_t)slot * 512);
    memset(p.h, 0, sizeof *p.h);
    const u32 nbins = p.nbins;
    const size_t per = (size_t)nbins + 8;
    DALLOC(c, p.d_bounds, u64 *, per * 8 * AG_BATCH);
    DALLOC(c, p.d_cnt, u64 *, per * 8 * AG_BATCH);
    DALLOC(c, p.d_off, u64 *, per * 8 * AG_BATCH);          // (the counts stay as they are: later rungs of the ladder fill in their bins and the scan runs again)
    DALLOC(c, p.d_flags, u32 *, 256);
    HIPCHK(c, hipMemsetAsync(p.d_flags, 0, 256, c->stream));
    if (!p.big) for (int x = 0; x < 2; ++x) DALLOC(c, p.d_list[x], u32 *, (size_t)nbins * 4 * AG_BATCH);
    AggArgs &a = p.a; memset(&a, 0, sizeof a);
    a.lower = L; a.upper = (u32)c->cfg.upper_freq; a.nbins = nbins; a.shift = 64 - prefix_bits; a.nw = NW;
    a.top_bits = (NW >= 2 && prefix_bits == 16) ? prefix_top_bits(K, NW) : 0;
    a.top_sig = NW >= 2 ? 2 * (K - 32 * (NW - 1)) : 64;
    u64 nmax = 0;
    for (int i = 0; i < AG_BATCH; ++i) {
        AggTask &t = a.t[i];
        p.bt[i] = bt[i];
        if (bt[i].n == 0) continue;
        u64 *other = (bt[i].out_k == bt[i].kA) ? bt[i].kB : bt[i].kA;
        t.keys = bt[i].out_k; t.vals = weighted ? bt[i].out_v : nullptr; t.n = bt[i].n; t.bounds = p.d_bounds + per * i; t.bin_cnt = p.d_cnt + per * i; t.bin_off = p.d_off + per * i; t.flags = p.d_flags + i;
        t.slot_shift = p.slot_shift; t.active = 1; p.ntot += bt[i].n; nmax = std::max(nmax, bt[i].n);
        if (!p.big) { t.ovf_list = p.d_list[0] + (size_t)nbins * i; t.ovf_n = p.d_flags + 2 * AG_BATCH + i; }      // first rung: all bins, overflowing ones listed
        if (p.slot_shift) t.scratch = other;             // the idle ping-pong buffer: n / 2 entries
        else {
            t.scratch = (u64 *)c->pool.alloc(bt[i].n * EW * 8 + 64); p.own_scratch[i] = true;
            if (!t.scratch) return fail(c, HSK_ERR_OOM, "finish scratch of %llu bytes", (unsigned long long)(bt[i].n * EW * 8));
        }
    }
    hipLaunchKernelGGL(bin_bounds_kernel, dim3(nbins / AG_THREADS + 1, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    if constexpr (NW <= 2) if (!weighted && !p.big && tune("agg_large", 1) != 0 && nmax >= AG_LARGE_BIN && (NW == 1 || a.top_bits == 0 || a.top_bits == 16)) {
        // bins of very many records (one k-mer seen millions of times): found, cut into slices and counted by many workgroups before the ladder
        // starts (hsk_agg.h: AggLarge); without such bins the two launches end at once
        size_t off[AG_BATCH][7], total = (sizeof(AggLarge) * AG_BATCH + 255) / 256 * 256;
        for (int i = 0; i < AG_BATCH; ++i) {
            const size_t sz[7] = {(size_t)nbins * 4, (size_t)AGL_TABLES * AGL_TAB * 8, (size_t)AGL_TABLES * AGL_TAB * 4, 256, (size_t)(bt[i].n / AGL_SLICE + AGL_TABLES + 1) * 8, 256,
                                  NW == 2 ? (size_t)AGL_TABLES * AGL_TAB * 8 : 0};
            for (int q = 0; q < 7; ++q) { off[i][q] = total; total += (sz[q] + 255) / 256 * 256; }
        }
        DALLOC(c, p.d_large, char *, total + 64);
        HIPCHK(c, hipMemsetAsync(p.d_large, 0, total, c->stream));
        AggLarge *h_lg = (AggLarge *)((char *)c->pinned + (512u << 10) + (size_t)slot * 1024);
        for (int i = 0; i < AG_BATCH; ++i) {
            HIPCHK(c, hipMemsetAsync(p.d_large + off[i][1], 0xFF, (size_t)AGL_TABLES * AGL_TAB * 8, c->stream));
            if (NW == 2) HIPCHK(c, hipMemsetAsync(p.d_large + off[i][6], 0xFF, (size_t)AGL_TABLES * AGL_TAB * 8, c->stream));
            AggLarge &g = h_lg[i];
            g.tkeys0 = NW == 2 ? (unsigned long long *)(p.d_large + off[i][6]) : nullptr;
            g.bin_tab = (u32 *)(p.d_large + off[i][0]); g.tkeys = (unsigned long long *)(p.d_large + off[i][1]); g.tcnt = (u32 *)(p.d_large + off[i][2]);
            g.tbad = (u32 *)(p.d_large + off[i][3]); g.units = (unsigned long long *)(p.d_large + off[i][4]); g.ctl = (u32 *)(p.d_large + off[i][5]);
            a.t[i].lg = (const AggLarge *)p.d_large + i;
        }
        HIPCHK(c, hipMemcpyAsync(p.d_large, h_lg, sizeof(AggLarge) * AG_BATCH, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(agg_large_list_kernel, dim3(nbins / AG_THREADS + 1, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        if (NW == 1) hipLaunchKernelGGL(agg_large_slice_kernel, dim3(160, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL(agg2_large_slice_kernel, dim3(160, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    }
    // First table: what the bins of the previous batches needed (hsk_ctx::agg_first_cap: error-free reads at ~30x stay on 1024
    // slots, reads with ~1 % errors move to 2048 after their first batch); bins of 6144 records and more on average (tasks far
    // above 2^28 k-mers) start on the large table.  Two-word keys: small / large only.
    p.first_cap = p.big ? AG_LOG2CAP_SMALL : std::max(c->agg_first_cap, nmax / nbins >= 6144 ? AG_LOG2CAP_LARGE : AG_LOG2CAP_SMALL);
    if (weighted) {                                      // pairs are (nearly) all distinct: the table that takes the average bin twice over
        const u64 avg = nmax / nbins + 1;
        p.first_cap = avg * 2 <= 600 ? AG_LOG2CAP_SMALL : avg * 2 <= 1200 ? AG_LOG2CAP_MEDIUM : AG_LOG2CAP_LARGE;
    }
    int rc = agg_launch_rung<NW>(c, p, p.first_cap, nbins, p.ntot); if (rc) return rc;
    rc = agg_launch_scan(c, p); if (rc) return rc;
    p.ev = ev_get(c);
    HIPCHK(c, hipEventRecord(p.ev, c->stream));
    p.active = true;
    return HSK_OK;
}

// {k-mer, count} records ordered by key (a key may occur several times: partial pairs of the combining extraction, the per-rank lists of a
// heavy-hitter task): counts of equal keys summed, [L, U] applied, entries and histogram out (hsk_heavy.h: heavy_merge_kernel)
template <int NW>
static int merge_sorted_pairs(hsk_ctx *c, const u64 *sk, const u64 *sv, u64 n, u64 *d_histo, u32 histo_len, TaskOut &out)
{
    out = TaskOut();
    if (n == 0) return HSK_OK;
    const u64 ntiles = (n + HV_THREADS - 1) / HV_THREADS;
    u64 *d_tile, *d_total;
    DALLOC(c, d_tile, u64 *, ntiles * 8 + 64); DALLOC(c, d_total, u64 *, 256);
    HeavyMergeArgs a; memset(&a, 0, sizeof a);
    a.keys = sk; a.cnts = sv; a.n = n; a.lower = (u64)c->cfg.lower_freq; a.upper = (u64)c->cfg.upper_freq; a.tile_cnt = d_tile; a.histo = d_histo; a.histo_len = histo_len;
    hipLaunchKernelGGL((heavy_merge_kernel<NW, false>), dim3((u32)ntiles), dim3(HV_THREADS), 0, c->stream, a);
    hipLaunchKernelGGL(count_scan_kernel, dim3(1), dim3(CNT_THREADS), 0, c->stream, d_tile, ntiles, d_total);
    u64 *tot = (u64 *)((char *)c->pinned + c->pinned_bytes - 128);
    HIPCHK(c, hipMemcpyAsync(tot, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));
    out.n = tot[0];
    if (out.n) {
        DALLOC(c, out.entries, u64 *, out.n * (NW + 1) * 8);
        a.entries = out.entries;
        hipLaunchKernelGGL((heavy_merge_kernel<NW, true>), dim3((u32)ntiles), dim3(HV_THREADS), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    c->pool.release(d_tile); c->pool.release(d_total);      // (stream-ordered reuse)
    return HSK_OK;
}

// covered: later work has already been enqueued behind stage 1 (the wait does not leave the GPU idle)
template <int NW>
static int agg_stage2(hsk_ctx *c, AggPending &p, u64 *d_histo, u32 histo_len, TaskOut *outs, bool covered)
{
    constexpr u32 EW = NW + 1;
    for (int i = 0; i < AG_BATCH; ++i) outs[i] = TaskOut();
    if (!p.active) return HSK_OK;
    AggArgs &a = p.a; const BatchTask *bt = p.bt; const u32 nbins = p.nbins; const bool big = p.big;
    c->stats.host_syncs++; if (covered) c->stats.host_waits_covered++;
    HIPCHK(c, hipEventSynchronize(p.ev));
    ev_put(c, p.ev); p.ev = nullptr;
    AggHostRead &h = *p.h;
    bool done[AG_BATCH];
    int rc = HSK_OK, nact = 0;
    u64 ovf_bins = 0;
    for (int i = 0; i < AG_BATCH; ++i) { done[i] = bt[i].n == 0 || !h.flags[i]; if (bt[i].n) { ++nact; ovf_bins += h.ovf[0][i]; } }
    if (!big && !c->forbid_long_way && !p.weighted && nact) {
        // next batch (and next call): one table size up when more than one bin in twenty did not fit this one (each of them is
        // read twice), one size down again when nothing overflowed and no bin came anywhere near this size's limit (one-word
        // keys: the kernel reports its fullest bin; multi-word keys: after four batches without an overflow)
        u32 maxd = 0; for (int i = 0; i < AG_BATCH; ++i) if (bt[i].n) maxd = std::max(maxd, h.maxd[i]);
        if (ovf_bins * 20 > (u64)nact * nbins) { c->agg_first_cap = std::min(p.first_cap + 1, (int)AG_LOG2CAP_LARGE); c->agg_clean_batches = 0; }
        else if (ovf_bins == 0 && p.first_cap > AG_LOG2CAP_SMALL &&
                 (NW == 1 ? maxd < (1u << (p.first_cap - 1)) * 3 / 4 : ++c->agg_clean_batches >= 4)) { c->agg_first_cap = p.first_cap - 1; c->agg_clean_batches = 0; }
    }
    { const bool force_off = tune("agg_adapt", 1) == 2;     // (tests: as if this batch had been found hopeless, but finished normally)
      if (force_off && !c->forbid_long_way && !p.weighted) { if (NW == 1) c->agg_off = true; else c->agg_off_wide = true; } }
    // ---- the ladder, bin by bin: the listed bins again one table size up, until no bin is left or the rungs are ----------------
    if (!big && ovf_bins) {
        AggArgs keep = a;
        int cap = p.first_cap, cur = 0;
        u32 n_cur[AG_BATCH]; for (int i = 0; i < AG_BATCH; ++i) n_cur[i] = bt[i].n ? h.ovf[0][i] : 0;
        for (;;) {
            u32 longest = 0; u64 nb = 0; for (int i = 0; i < AG_BATCH; ++i) { longest = std::max(longest, n_cur[i]); nb += n_cur[i]; }
            if (!longest) break;
            const int max_rung = (int)tune("agg_maxrung", AG_LOG2CAP_HUGE);     // (tests: stop the ladder early, the listed bins' tasks take the long way)
            int next = (NW >= 2 || p.weighted) ? (cap < AG_LOG2CAP_LARGE ? cap + 1 : 0) : (cap < AG_LOG2CAP_HUGE ? cap + 1 : 0);
            if (next > max_rung) next = 0;
            if (!next) { for (int i = 0; i < AG_BATCH; ++i) if (n_cur[i]) done[i] = false; break; }   // a bin beyond the last rung: the task takes the long way
            // More than half of all bins did not fit 2048 slots: this input has (nearly) as many distinct k-mers as k-mers -- reads with
            // 5 % errors and more -- and the aggregation is the wrong tool.  No further rungs: the tasks of this batch take the long way
            // now, the batches after it (and later calls on this context) four prefix passes + the tile finish instead of two + tables.
            const bool adapt = tune("agg_adapt", 1) != 0;
            // (multi-word keys have no tile finish to change to: their tasks just stop climbing a ladder that ends in the long way anyway)
            if (adapt && !c->forbid_long_way && !p.weighted && cap >= AG_LOG2CAP_MEDIUM && nb * 2 > (u64)nact * nbins) {
                if (NW == 1) c->agg_off = true; else c->agg_off_wide = true;
                for (int i = 0; i < AG_BATCH; ++i) if (n_cur[i]) done[i] = false;
                break;
            }
            cap = next;
            const bool last = (NW >= 2 || p.weighted) ? cap == AG_LOG2CAP_LARGE : cap == AG_LOG2CAP_HUGE;
            HIPCHK(c, hipMemsetAsync(p.d_flags + (2 + (cur ^ 1)) * AG_BATCH, 0, sizeof(u32) * AG_BATCH, c->stream));
            for (int i = 0; i < AG_BATCH; ++i) {
                AggTask &t = a.t[i];
                t.active = (keep.t[i].active && n_cur[i]) ? 1 : 0;
                if (t.active) c->stats.agg_retried_tasks++;
                t.bin_list = p.d_list[cur] + (size_t)nbins * i; t.bin_list_n = p.d_flags + (2 + cur) * AG_BATCH + i;
                t.ovf_list = last ? nullptr : p.d_list[cur ^ 1] + (size_t)nbins * i; t.ovf_n = p.d_flags + (2 + (cur ^ 1)) * AG_BATCH + i;
            }
            rc = agg_launch_rung<NW>(c, p, cap, longest, nb * (p.ntot / ((u64)nact * nbins) + 1)); if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(p.h->flags, p.d_flags, 4 * sizeof(u32) * AG_BATCH, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hsk_sync(c, c->stream));
            if (last) { for (int i = 0; i < AG_BATCH; ++i) if (a.t[i].active && h.flags[i]) done[i] = false; break; }
            cur ^= 1;
            for (int i = 0; i < AG_BATCH; ++i) n_cur[i] = a.t[i].active ? h.ovf[cur][i] : 0;
        }
        a = keep;
        for (int i = 0; i < AG_BATCH; ++i) { a.t[i].bin_list = nullptr; a.t[i].bin_list_n = nullptr; }
        rc = agg_launch_scan(c, p); if (rc) return rc;        // the bin counts are complete now: offsets and totals again
        HIPCHK(c, hsk_sync(c, c->stream));
    }
    u64 total[AG_BATCH];
    for (int i = 0; i < AG_BATCH; ++i) total[i] = h.total[i];
    AggCompactArgs ca; memset(&ca, 0, sizeof ca);
    ca.slot_shift = p.slot_shift; ca.histo = d_histo; ca.histo_len = histo_len; ca.nbins = nbins; ca.ew = EW;
    bool any = false;
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || !done[i]) continue;
        c->stats.fused_tasks++;
        outs[i].n = total[i];
        if (outs[i].n) {
            outs[i].entries = (u64 *)c->pool.alloc(outs[i].n * EW * 8);
            if (!outs[i].entries) { rc = fail(c, HSK_ERR_OOM, "task output of %llu bytes", (unsigned long long)(outs[i].n * EW * 8)); break; }
            ca.scratch[i] = a.t[i].scratch; ca.bounds[i] = a.t[i].bounds; ca.bin_off[i] = a.t[i].bin_off; ca.entries[i] = outs[i].entries;
            any = true;
        }
    }
    if (any && rc == HSK_OK) hipLaunchKernelGGL(agg_compact_kernel, dim3(big ? 64 : 256, AG_BATCH), dim3(AG_THREADS), 0, c->stream, ca);
    if (!big && !c->forbid_long_way && !p.weighted && rc == HSK_OK) {
        bool redo[AG_BATCH]; int nredo = 0;
        for (int i = 0; i < AG_BATCH; ++i) { redo[i] = bt[i].n != 0 && !done[i]; nredo += redo[i]; }
        if (nredo >= 3) {
            for (int i = 0; i < AG_BATCH; ++i) if (redo[i] && p.own_scratch[i]) { c->pool.release(a.t[i].scratch); a.t[i].scratch = nullptr; p.own_scratch[i] = false; }   // (stream-ordered reuse)
            rc = long_way_batch<NW>(c, bt, redo, p.K, nullptr, d_histo, histo_len, outs);
            for (int i = 0; i < AG_BATCH; ++i) if (redo[i]) done[i] = true;
        }
    }
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || done[i]) continue;
        if (big || c->forbid_long_way) { outs[i].failed = true; continue; }
        if (p.weighted) {
            // {k-mer, count} pairs with a bin beyond the last table (a skewed key prefix: poly-A, satellites): the long way for this task --
            // full-width passes over the pairs with the counts as payload, then equal keys summed (round 4; rounds 2-3 ran the whole CALL
            // again on the instance path, which several ranks cannot do: their peers would wait in the exchange)
            c->stats.redone_tasks++;
            if (p.own_scratch[i]) { c->pool.release(a.t[i].scratch); a.t[i].scratch = nullptr; p.own_scratch[i] = false; }
            SortScratch scw; rc = alloc_sort_scratch(c, scw); if (rc) break;
            u64 *ck = bt[i].out_k, *ok_ = (ck == bt[i].kA) ? bt[i].kB : bt[i].kA, *cv = bt[i].out_v, *ov = (cv == bt[i].vA) ? bt[i].vB : bt[i].vA, *sk, *sv;
            rc = sort_task_device<NW>(c, ck, ok_, cv, ov, bt[i].n, p.K, scw, &sk, &sv, false);
            free_sort_scratch(c, scw);
            if (rc == HSK_OK) rc = merge_sorted_pairs<NW>(c, sk, sv, bt[i].n, d_histo, histo_len, outs[i]);
            continue;
        }
        // the long way for this task: full-width passes from the current order, then the two-pass counter
        c->stats.redone_tasks++;
        if (p.own_scratch[i]) { c->pool.release(a.t[i].scratch); a.t[i].scratch = nullptr; p.own_scratch[i] = false; }   // (stream-ordered reuse)
        SortScratch sc1; rc = alloc_sort_scratch(c, sc1); if (rc) break;
        u64 *cur = bt[i].out_k, *other = (cur == bt[i].kA) ? bt[i].kB : bt[i].kA, *sk, *sv;
        rc = sort_task_device<NW>(c, cur, other, nullptr, nullptr, bt[i].n, p.K, sc1, &sk, &sv, false);
        free_sort_scratch(c, sc1);
        if (rc == HSK_OK) rc = count_task_device<NW>(c, sk, nullptr, bt[i].n, 0, d_histo, histo_len, outs[i]);
    }
    HIPCHK(c, hipGetLastError());
    // no wait here: the scratch (the batch's idle ping-pong buffers, or pool blocks) is next touched by work that is
    // enqueued on this stream after the compaction
    for (int i = 0; i < AG_BATCH; ++i) if (p.own_scratch[i]) c->pool.release(a.t[i].scratch);
    c->pool.release(p.d_bounds); c->pool.release(p.d_cnt); c->pool.release(p.d_off); c->pool.release(p.d_flags); c->pool.release(p.d_list[0]); c->pool.release(p.d_list[1]); c->pool.release(p.d_large);
    p.active = false;
    return rc;
}

// ---- EXTENSION: two passes + grouping aggregation (hsk_agg.h: agg_ext_kernel) ---------------------------------------
// pay_before[i]: offset of task i's payload range in the rank's payload arrays (payload_off values are global over the
// owned tasks in ascending id).
template <int NW>
static int agg_ext_finish_batch_device(hsk_ctx *c, BatchTask *bt, int K, const u64 *pay_before, u64 *d_histo, u32 histo_len, TaskOut *outs)
{
    static_assert(NW <= 3, "keys of one to three words");
    constexpr u32 EW = NW + 1;                          // words per entry
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    const u32 L = (u32)c->cfg.lower_freq;
    const u32 slot_shift = L >= 2 ? 1 : 0;
    const u32 nbins = AG_BINS;
    const size_t per = (size_t)nbins + 8;
    u64 *d_bounds, *d_cnt; u32 *d_flags;
    DALLOC(c, d_bounds, u64 *, per * 8 * AG_BATCH);
    DALLOC(c, d_cnt, u64 *, per * 8 * AG_BATCH);
    DALLOC(c, d_flags, u32 *, 256);
    HIPCHK(c, hipMemsetAsync(d_flags, 0, 64, c->stream));
    AggExtArgs a; memset(&a, 0, sizeof a);
    AggArgs sa; memset(&sa, 0, sizeof sa);               // the view agg_scan_kernel needs
    a.lower = L; a.upper = (u32)c->cfg.upper_freq; a.nbins = nbins; a.shift = AG_SHIFT; a.nw = NW; a.top_bits = NW >= 2 ? prefix_top_bits(K, NW) : 0; a.top_sig = NW >= 2 ? 2 * (K - 32 * (NW - 1)) : 64;
    sa.nbins = nbins; sa.shift = AG_SHIFT; sa.nw = NW;
    bool own_scratch[AG_BATCH] = {false};
    u64 ntot = 0;
    for (int i = 0; i < AG_BATCH; ++i) {
        AggExtTask &t = a.t[i];
        outs[i] = TaskOut();
        if (bt[i].n == 0) continue;
        u64 *other_k = (bt[i].out_k == bt[i].kA) ? bt[i].kB : bt[i].kA;
        u64 *other_v = (bt[i].out_v == bt[i].vA) ? bt[i].vB : bt[i].vA;
        t.keys = bt[i].out_k; t.vals = bt[i].out_v; t.n = bt[i].n; t.bounds = d_bounds + per * i; t.bin_cnt = d_cnt + per * i; t.flags = d_flags + i;
        t.slot_shift = slot_shift; t.active = 1; t.payoff_add = pay_before[i]; ntot += bt[i].n;
        if (slot_shift) { t.scratch_e = other_k; t.scratch_p = other_v; }       // n / 2 entries of 8 (NW + 1) + 8 bytes: the idle ping-pong buffers (8 NW + 8 bytes per record)
        else {
            t.scratch_e = (u64 *)c->pool.alloc(bt[i].n * EW * 8 + 64); t.scratch_p = (u64 *)c->pool.alloc(bt[i].n * 8 + 64); own_scratch[i] = true;
            if (!t.scratch_e || !t.scratch_p) return fail(c, HSK_ERR_OOM, "finish scratch");
        }
        outs[i].npay = bt[i].n;
        DALLOC(c, outs[i].pos, u32 *, bt[i].n * 4); DALLOC(c, outs[i].rid, int32_t *, bt[i].n * 4);
        t.pos = outs[i].pos; t.rid = outs[i].rid;
        sa.t[i].bin_cnt = t.bin_cnt; sa.t[i].bin_off = t.bin_cnt; sa.t[i].active = 1;       // (in place: this path scans once)
    }
    // The table ladder, bin by bin (as agg_stage1 / agg_stage2 do for keys without payload: nearly every task has a few outlier
    // bins, a task-wide retry would run everything on the largest table's one workgroup per CU): the first table over all bins,
    // the bins it could not hold listed; the next table over the listed bins, and so on; a bin beyond the last table sends its
    // task the long way.  14 + 16 NW bytes of LDS per slot: 1024 / 2048 / 4096 slots for one-word keys (5 / 2 / 1 workgroups per
    // CU), 1024 / 2048 for two and three words.  The first table follows the previous batches like agg_stage2's (reads with
    // ~1 % errors: ~900 distinct keys per bin, every bin overflows 1024 slots).
    constexpr int TOP = NW == 1 ? AG_LOG2CAP_LARGE : AG_LOG2CAP_MEDIUM;
    u32 *d_list; DALLOC(c, d_list, u32 *, (size_t)nbins * 4 * AG_BATCH * 2);
    struct { u32 flags[3 * AG_BATCH]; u64 total[AG_BATCH]; } h;
    auto launch = [&](int log2cap, u32 grid_x) {
        EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 2; ep.keys = ntot; ep.bytes = ntot * 16; (void)hipEventRecord(ep.a, c->stream); }
        if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg_ext_kernel<AG_LOG2CAP_SMALL, NW>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else if (log2cap == AG_LOG2CAP_MEDIUM || NW > 1) hipLaunchKernelGGL((agg_ext_kernel<AG_LOG2CAP_MEDIUM, NW>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((agg_ext_kernel<(NW == 1 ? AG_LOG2CAP_LARGE : AG_LOG2CAP_MEDIUM), NW>), dim3(grid_x, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
    };
    memset(&h, 0, sizeof h);
    hipLaunchKernelGGL(bin_bounds_ext_kernel, dim3(nbins / AG_THREADS + 1, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    int nact = 0; for (int i = 0; i < AG_BATCH; ++i) nact += a.t[i].active ? 1 : 0;
    bool hopeless[AG_BATCH] = {false};
    int cap = std::min(std::max(c->agg_first_cap, (int)AG_LOG2CAP_SMALL), TOP - 1);     // (never the last table first: its overflows are not counted)
    const int first_cap = cap;
    {
        const AggExtArgs keep = a;
        int cur = 0;                                        // list the running rung appends to
        u32 grid_x = nbins;
        for (bool first = true;; first = false) {
            const bool last = cap == TOP;
            for (int i = 0; i < AG_BATCH; ++i) {
                AggExtTask &t = a.t[i];
                if (!first) { t.bin_list = d_list + ((size_t)(cur ^ 1) * AG_BATCH + i) * nbins; t.bin_list_n = d_flags + (1 + (cur ^ 1)) * AG_BATCH + i; }
                t.ovf_list = last ? nullptr : d_list + ((size_t)cur * AG_BATCH + i) * nbins;
                t.ovf_n = last ? nullptr : d_flags + (1 + cur) * AG_BATCH + i;
            }
            if (!last) HIPCHK(c, hipMemsetAsync(d_flags + (1 + cur) * AG_BATCH, 0, sizeof(u32) * AG_BATCH, c->stream));
            launch(cap, grid_x);
            HIPCHK(c, hipGetLastError());
            if (last) break;
            HIPCHK(c, hipMemcpyAsync(h.flags, d_flags, sizeof h.flags, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hsk_sync(c, c->stream));
            u32 longest = 0; u64 listed = 0;
            for (int i = 0; i < AG_BATCH; ++i) if (keep.t[i].active) { const u32 n = std::min(h.flags[(1 + cur) * AG_BATCH + i], nbins); longest = std::max(longest, n); listed += n; }
            if (first && nact) {                            // the next batch's first table
                if (listed * 20 > (u64)nact * nbins) { c->agg_first_cap = std::min(first_cap + 1, TOP - 1); c->agg_clean_batches = 0; }
                else if (listed == 0 && first_cap > AG_LOG2CAP_SMALL && ++c->agg_clean_batches >= 4) { c->agg_first_cap = first_cap - 1; c->agg_clean_batches = 0; }
            }
            if (!longest) break;
            // more than half of all bins beyond 2048 slots: (nearly) as many distinct k-mers as k-mers; the listed bins' tasks take the
            // long way now instead of after the last table
            const bool adapt = tune("agg_adapt", 1) != 0;
            if (adapt && cap >= AG_LOG2CAP_MEDIUM && listed * 2 > (u64)nact * nbins) {
                for (int i = 0; i < AG_BATCH; ++i) if (keep.t[i].active && h.flags[(1 + cur) * AG_BATCH + i]) hopeless[i] = true;
                c->agg_off_wide = true;                      // the batches after this one: no prefix passes and tables at all
                break;
            }
            for (int i = 0; i < AG_BATCH; ++i) { a.t[i].active = (keep.t[i].active && h.flags[(1 + cur) * AG_BATCH + i]) ? 1 : 0; c->stats.agg_retried_tasks += a.t[i].active; }
            ++cap; cur ^= 1; grid_x = longest;
        }
        a = keep;
    }
    { const bool force_off = tune("agg_adapt", 1) == 2;     // (tests: as in agg_stage2)
      if (force_off) c->agg_off_wide = true; }
    for (int i = 0; i < AG_BATCH; ++i) sa.t[i].active = a.t[i].active;
    hipLaunchKernelGGL(agg_scan_kernel, dim3(AG_BATCH), dim3(AG_THREADS), 0, c->stream, sa);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h.flags, d_flags, sizeof h.flags, hipMemcpyDeviceToHost, c->stream));
    for (int i = 0; i < AG_BATCH; ++i) if (a.t[i].active) HIPCHK(c, hipMemcpyAsync(&h.total[i], a.t[i].bin_cnt + nbins, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));
    int rc = HSK_OK;
    bool done[AG_BATCH];
    u64 total[AG_BATCH];
    for (int i = 0; i < AG_BATCH; ++i) { done[i] = bt[i].n == 0 || (!h.flags[i] && !hopeless[i]); total[i] = h.total[i]; }
    AggExtCompactArgs ca; memset(&ca, 0, sizeof ca);
    ca.slot_shift = slot_shift; ca.histo = d_histo; ca.histo_len = histo_len; ca.nbins = nbins; ca.ew = EW;
    bool any = false;
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || !done[i]) continue;
        c->stats.fused_tasks++;
        outs[i].n = total[i];
        if (outs[i].n) {
            outs[i].entries = (u64 *)c->pool.alloc(outs[i].n * EW * 8); outs[i].payoff = (u64 *)c->pool.alloc(outs[i].n * 8);
            if (!outs[i].entries || !outs[i].payoff) { rc = fail(c, HSK_ERR_OOM, "task output"); break; }
            ca.scratch_e[i] = a.t[i].scratch_e; ca.scratch_p[i] = a.t[i].scratch_p; ca.bounds[i] = a.t[i].bounds; ca.bin_off[i] = a.t[i].bin_cnt;
            ca.entries[i] = outs[i].entries; ca.payoff[i] = outs[i].payoff;
            any = true;
        }
    }
    if (any && rc == HSK_OK) hipLaunchKernelGGL(agg_ext_compact_kernel, dim3(256, AG_BATCH), dim3(AG_THREADS), 0, c->stream, ca);
    {
        bool redo[AG_BATCH]; int nredo = 0;
        for (int i = 0; i < AG_BATCH; ++i) { redo[i] = bt[i].n != 0 && !done[i]; nredo += redo[i]; }
        if (nredo >= 3 && rc == HSK_OK) {
            HIPCHK(c, hsk_sync(c, c->stream));
            for (int i = 0; i < AG_BATCH; ++i) if (redo[i]) {
                if (own_scratch[i]) { c->pool.release(a.t[i].scratch_e); c->pool.release(a.t[i].scratch_p); own_scratch[i] = false; }
                free_task_out(c, outs[i]);
            }
            rc = long_way_batch<NW>(c, bt, redo, K, pay_before, d_histo, histo_len, outs);
            for (int i = 0; i < AG_BATCH; ++i) if (redo[i]) done[i] = true;
        }
    }
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || done[i]) continue;
        // the long way for this task: full-width passes (payload carried) from the current order, then the two-pass counter
        c->stats.redone_tasks++;
        HIPCHK(c, hsk_sync(c, c->stream));
        if (own_scratch[i]) { c->pool.release(a.t[i].scratch_e); c->pool.release(a.t[i].scratch_p); own_scratch[i] = false; }
        const u64 payadd = pay_before[i];
        free_task_out(c, outs[i]);
        SortScratch sc1; rc = alloc_sort_scratch(c, sc1); if (rc) break;
        u64 *cur = bt[i].out_k, *other = (cur == bt[i].kA) ? bt[i].kB : bt[i].kA, *sk, *sv;
        u64 *vcur = bt[i].out_v, *vother = (vcur == bt[i].vA) ? bt[i].vB : bt[i].vA;
        rc = sort_task_device<NW>(c, cur, other, vcur, vother, bt[i].n, K, sc1, &sk, &sv, false);
        free_sort_scratch(c, sc1);
        if (rc == HSK_OK) rc = count_task_device<NW>(c, sk, sv, bt[i].n, payadd, d_histo, histo_len, outs[i]);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hsk_sync(c, c->stream));
    for (int i = 0; i < AG_BATCH; ++i) if (own_scratch[i]) { c->pool.release(a.t[i].scratch_e); c->pool.release(a.t[i].scratch_p); }
    c->pool.release(d_bounds); c->pool.release(d_cnt); c->pool.release(d_flags); c->pool.release(d_list);
    return rc;
}
