// hsk_host_sort.h -- host side of the sort stage: digit plans, single-task / eight-task / many-task scatter passes (kernels: hsk_sort.h).
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// ------------------------------------------------------------------------------------------------
// stage: sort one task (a12)
// ------------------------------------------------------------------------------------------------
// Digit plan: the key is the little-endian integer formed by words 0..NW-1; word w carries
// min(32, K-32w) bases in its top bits.  Digits are taken from the least significant used bit up.
static int make_pass_plan(int K, int nw, int rb, PassDesc *out)
{
    int np = 0;
    for (int w = 0; w < nw; ++w) {
        const int nbases = std::min(32, K - 32 * w);
        int lo = 64 - 2 * nbases;
        while (lo < 64) { int bits = std::min(rb, 64 - lo); out[np++] = PassDesc{w, lo, bits}; lo += bits; }
    }
    return np;
}

// Hybrid plan (one-word keys without payload): only the top 32 bits (16 bases) are ordered by global passes
// (digits at bit 32, 40, 48, 56, least significant first); binsort_kernel finishes the low bits inside each
// bin.  With 32 prefix bits two different k-mers of one task rarely share a bin, so nearly every bin is the
// copies of ONE k-mer and passes through untouched; 24 bits left 40 % of the records in multi-key bins whose
// in-LDS ordering (serial, LDS-latency bound) cost more than the fourth pass.
constexpr int HYBRID_SHIFT = 32;
static int make_hybrid_plan(PassDesc *out, int prefix_bits = 64 - HYBRID_SHIFT, int word = 0)
{
    const int np = prefix_bits / 8;                     // LSD passes over the top prefix_bits bits (of the most significant word)
    for (int i = 0; i < np; ++i) out[i] = PassDesc{word, 64 - prefix_bits + 8 * i, 8};
    return np;
}
static bool hybrid_enabled()
{
    return !(g_plan_flags & HSK_FLAG_FULL_SORT);
}
// Two- and three-word keys take the prefix plan when the aggregating finish follows.  Where the most significant word carries
// fewer than the 16 prefix bits (K - 32 (NW - 1) < 8 bases), the prefix continues in the top bits of the word below it
// (make_split_prefix_plan).
template <int NW> static bool prefix_plan_ok(int K, bool finish_follows)
{
    return hybrid_enabled() && (NW == 1 || (NW <= 3 && finish_follows));
}
// bits of the 16-bit prefix that the most significant word holds (16: all of them)
static int prefix_top_bits(int K, int nw) { return std::min(16, 2 * (K - 32 * (nw - 1))); }
// The 16-bit prefix of a key whose most significant word has only `top` < 16 significant bits: those, then the top 16 - top
// bits of the word below.  LSD passes, least significant digit first, no digit across a word boundary or wider than 8 bits.
static int make_split_prefix_plan(PassDesc *out, int top, int nw)
{
    int np = 0;
    const int low = 16 - top;                            // bits taken from word nw - 2
    for (int lo = 64 - low; lo < 64; ) { const int bits = std::min(8, 64 - lo); out[np++] = PassDesc{nw - 2, lo, bits}; lo += bits; }
    for (int lo = 64 - top; lo < 64; ) { const int bits = std::min(8, 64 - lo); out[np++] = PassDesc{nw - 1, lo, bits}; lo += bits; }
    return np;
}

// tasks of 2^30 keys and more use 64-bit look-back words; HSK_WIDE_LOOKBACK=1 forces them (tests: such tasks do not fit a test)
static bool force_wide_lookback()
{
    return tune("wide_lookback", 0) != 0;
}

// The first of several prefix passes need not be stable when an aggregation (which only needs the records GROUPED by the
// prefix) follows: its ranking is then a single LDS atomic per key.  HSK_UNSTABLE_FIRST=0 keeps it stable.
static bool unstable_first_pass()
{
    return tune("unstable_first", 1) != 0;
}

struct SortScratch {
    u64 *ghist = nullptr;      // [MAX_PASSES][256]
    u64 *gbase = nullptr;      // [MAX_PASSES][256]
    void *lookback = nullptr; size_t lookback_bytes = 0;
    u32 *tickets = nullptr;    // [MAX_PASSES]
};

template <int NW, bool HAS_VAL, typename LB>
static void launch_onesweep(hsk_ctx *c, const SortArgs &a, u32 ntiles)
{
    hipLaunchKernelGGL((onesweep_kernel<NW, HAS_VAL, LB>), dim3(ntiles), dim3(SORT_THREADS), 0, c->stream, a);
}

// Sorts n records in bufA (keys) / valA using bufB / valB as the ping-pong buffer.  On return
// *out_keys / *out_vals point at whichever buffer holds the sorted data.
template <int NW>
static int sort_task_device(hsk_ctx *c, u64 *keysA, u64 *keysB, u64 *valsA, u64 *valsB, u64 n, int K, SortScratch &sc,
                            u64 **out_keys, u64 **out_vals, bool allow_hybrid = true)
{
    *out_keys = keysA; *out_vals = valsA;
    if (n < 2) return HSK_OK;
    const bool has_val = valsA != nullptr;
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    const bool hybrid = allow_hybrid && NW == 1 && hybrid_enabled();
    HistArgs h; memset(&h, 0, sizeof h);
    h.keys = keysA; h.n = n; h.npass = hybrid ? make_hybrid_plan(h.pass) : make_pass_plan(K, NW, c->cfg.radix_bits, h.pass); h.ghist = sc.ghist;
    HIPCHK(c, hipMemsetAsync(sc.ghist, 0, (size_t)MAX_PASSES * 256 * 8, c->stream));
    const u32 hblocks = (u32)std::min<u64>((n + SORT_THREADS * 16 - 1) / (SORT_THREADS * 16), 2048);
    EvPair hp{}; if (profile) { hp.a = ev_get(c); hp.b = ev_get(c); hp.kind = 1; hp.bytes = n * NW * 8; (void)hipEventRecord(hp.a, c->stream); }
    hipLaunchKernelGGL((hist_kernel<NW>), dim3(hblocks), dim3(SORT_THREADS), (size_t)h.npass * 256 * 4, c->stream, h);
    if (profile) { (void)hipEventRecord(hp.b, c->stream); c->ev_pending.push_back(hp); }
    u64 *hh = (u64 *)c->pinned;                          // [npass][256] histogram, then [npass][256] bases
    HIPCHK(c, hipMemcpyAsync(hh, sc.ghist, (size_t)h.npass * 256 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));
    u64 *hb = hh + (size_t)MAX_PASSES * 256;
    std::vector<int> todo;
    for (int p = 0; p < h.npass; ++p) {
        bool trivial = false; u64 run = 0;
        for (int d = 0; d < 256; ++d) { if (hh[p * 256 + d] == n) trivial = true; hb[p * 256 + d] = run; run += hh[p * 256 + d]; }
        if (!trivial) todo.push_back(p);
    }
    if (todo.empty() && !hybrid) return HSK_OK;
    u64 *kin = keysA, *kout = keysB, *vin = valsA, *vout = valsB;
    if (!todo.empty()) {
    HIPCHK(c, hipMemcpyAsync(sc.gbase, hb, (size_t)h.npass * 256 * 8, hipMemcpyHostToDevice, c->stream));
    constexpr int TILE = SortTile<NW>::TILE;
    const u32 ntiles = (u32)((n + TILE - 1) / TILE);
    const bool wide = n >= (1ULL << 30) || force_wide_lookback();
    const size_t lbw = wide ? 8 : 4;
    const size_t need = (size_t)todo.size() * ntiles * 256 * lbw;
    if (need > sc.lookback_bytes) {
        c->pool.release(sc.lookback);
        sc.lookback = c->pool.alloc(need); sc.lookback_bytes = need;
        if (!sc.lookback) { sc.lookback_bytes = 0; return fail(c, HSK_ERR_OOM, "look-back table of %zu bytes", need); }
    }
    HIPCHK(c, hipMemsetAsync(sc.lookback, 0, need, c->stream));
    HIPCHK(c, hipMemsetAsync(sc.tickets, 0, MAX_PASSES * 4, c->stream));
    for (size_t i = 0; i < todo.size(); ++i) {
        const int p = todo[i];
        SortArgs a; memset(&a, 0, sizeof a);
        a.keys_in = kin; a.keys_out = kout; a.vals_in = vin; a.vals_out = vout; a.n = n;
        a.word = h.pass[p].word; a.shift = h.pass[p].shift; a.bits = h.pass[p].bits;
        a.ntiles = ntiles;
        a.gbase = sc.gbase + (size_t)p * 256;
        a.lookback = (char *)sc.lookback + i * (size_t)ntiles * 256 * lbw;
        a.ticket = sc.tickets + i; a.err = c->d_err;
        EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 0; ep.keys = n; ep.bytes = 2 * n * (NW * 8 + (has_val ? 8 : 0)); (void)hipEventRecord(ep.a, c->stream); }
        if (has_val) { if (wide) launch_onesweep<NW, true, u64>(c, a, ntiles); else launch_onesweep<NW, true, u32>(c, a, ntiles); }
        else { if (wide) launch_onesweep<NW, false, u64>(c, a, ntiles); else launch_onesweep<NW, false, u32>(c, a, ntiles); }
        if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
        std::swap(kin, kout); std::swap(vin, vout);
    }
    HIPCHK(c, hipGetLastError());
    }
    if (hybrid) {
        // order the low bits inside every prefix bin (one more streaming pass instead of five scatter passes)
        u32 *d_flag = sc.tickets + 60;                         // spare word of the ticket block
        HIPCHK(c, hipMemsetAsync(d_flag, 0, 4, c->stream));
        BinSortArgs b; b.in = kin; b.out = kout; b.vin = vin; b.vout = vout; b.n = n; b.hi_shift = HYBRID_SHIFT; b.mixed_giant = d_flag;
        hipLaunchKernelGGL(binsort_kernel, dim3((u32)((n + BS_TILE - 1) / BS_TILE)), dim3(BS_THREADS), 0, c->stream, b);
        HIPCHK(c, hipGetLastError());
        std::swap(kin, kout); std::swap(vin, vout);
        u32 *hf = (u32 *)((char *)c->pinned + c->pinned_bytes - 192);
        HIPCHK(c, hipMemcpyAsync(hf, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hsk_sync(c, c->stream));
        if (*hf) {                                             // a long bin with several keys: finish with the full-width passes
            c->stats.redone_tasks++;
            u64 *other = (kin == keysA) ? keysB : keysA;
            u64 *vother = has_val ? ((vin == valsA) ? valsB : valsA) : nullptr;
            return sort_task_device<NW>(c, kin, other, vin, vother, n, K, sc, out_keys, out_vals, false);
        }
    }
    *out_keys = kin; *out_vals = vin;
    return HSK_OK;
}

static int alloc_sort_scratch(hsk_ctx *c, SortScratch &sc);
static void free_sort_scratch(hsk_ctx *c, SortScratch &sc);

// ---- eight tasks at a time, one per XCD (onesweep_multi_kernel) -----------------------------------------
struct BatchTask { u64 n = 0; u64 *kA = nullptr, *kB = nullptr, *vA = nullptr, *vB = nullptr; u64 *out_k = nullptr, *out_v = nullptr; };
constexpr int XCD_BATCH = 8;

template <int NW, bool HAS_VAL, typename LB>
static void launch_onesweep_multi(hsk_ctx *c, const MultiSortArgs &m, u32 grid)
{
    hipLaunchKernelGGL((onesweep_multi_kernel<NW, HAS_VAL, LB>), dim3(grid), dim3(SORT_THREADS), 0, c->stream, m);
}

// the digit plan of a batch sort (shared with expand_batch, which counts the digits while it writes the keys)
template <int NW>
static int batch_pass_plan(hsk_ctx *c, int K, bool finish_follows, int prefix_bits, PassDesc *plan)
{
    const bool hybrid = prefix_plan_ok<NW>(K, finish_follows);
    if (hybrid && finish_follows && NW >= 2 && prefix_top_bits(K, NW) < 16 && prefix_bits == 16) return make_split_prefix_plan(plan, prefix_top_bits(K, NW), NW);
    return hybrid ? make_hybrid_plan(plan, finish_follows ? prefix_bits : 64 - HYBRID_SHIFT, NW - 1) : make_pass_plan(K, NW, c->cfg.radix_bits, plan);
}

// d_ghist_pre: [XCD_BATCH][MAX_PASSES][256] digit histograms already counted by expand_batch (null: hist_kernel runs here)
template <int NW>
static int sort_batch_device(hsk_ctx *c, BatchTask *bt, int K, bool finish_follows, int prefix_bits = 64 - HYBRID_SHIFT, u64 *d_ghist_pre = nullptr)
{
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    const bool has_val = bt[0].vA != nullptr;
    constexpr int TILE = SortTile<NW>::TILE;
    u64 *d_ghist, *d_gbase; u32 *d_tickets;
    if (d_ghist_pre) d_ghist = d_ghist_pre;
    else {
        DALLOC(c, d_ghist, u64 *, (size_t)XCD_BATCH * MAX_PASSES * 256 * 8);
        HIPCHK(c, hipMemsetAsync(d_ghist, 0, (size_t)XCD_BATCH * MAX_PASSES * 256 * 8, c->stream));
    }
    DALLOC(c, d_gbase, u64 *, (size_t)XCD_BATCH * MAX_PASSES * 256 * 8);
    DALLOC(c, d_tickets, u32 *, (size_t)XCD_BATCH * MAX_PASSES * 4 + 256);       // + 8 flag words behind the tickets
    HIPCHK(c, hipMemsetAsync(d_tickets, 0, (size_t)XCD_BATCH * MAX_PASSES * 4 + 64, c->stream));
    PassDesc plan[MAX_PASSES];
    const bool hybrid = prefix_plan_ok<NW>(K, finish_follows);
    const int npass = batch_pass_plan<NW>(c, K, finish_follows, prefix_bits, plan);
    u64 ntot = 0; bool wide = force_wide_lookback();
    for (int i = 0; i < XCD_BATCH; ++i) {
        bt[i].out_k = bt[i].kA; bt[i].out_v = bt[i].vA;
        ntot += bt[i].n; if (bt[i].n >= (1ULL << 30)) wide = true;
        if (bt[i].n == 0 || d_ghist_pre) continue;
        HistArgs h; memset(&h, 0, sizeof h);
        h.keys = bt[i].kA; h.n = bt[i].n; h.npass = npass; memcpy(h.pass, plan, sizeof(PassDesc) * npass);
        h.ghist = d_ghist + (size_t)i * MAX_PASSES * 256;
        const u32 hblocks = (u32)std::min<u64>((bt[i].n + SORT_THREADS * 16 - 1) / (SORT_THREADS * 16), 2048);
        EvPair hp{}; if (profile) { hp.a = ev_get(c); hp.b = ev_get(c); hp.kind = 1; hp.bytes = bt[i].n * NW * 8; (void)hipEventRecord(hp.a, c->stream); }
        hipLaunchKernelGGL((hist_kernel<NW>), dim3(hblocks), dim3(SORT_THREADS), (size_t)npass * 256 * 4, c->stream, h);
        if (profile) { (void)hipEventRecord(hp.b, c->stream); c->ev_pending.push_back(hp); }
    }
    std::vector<u64> hh((size_t)XCD_BATCH * MAX_PASSES * 256), hb((size_t)XCD_BATCH * MAX_PASSES * 256, 0);
    HIPCHK(c, hipMemcpyAsync(hh.data(), d_ghist, hh.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));
    std::vector<int> todo;
    for (int p = 0; p < npass; ++p) {
        bool all_trivial = true;
        for (int i = 0; i < XCD_BATCH; ++i) {
            if (bt[i].n < 2) continue;
            bool trivial = false; u64 run = 0;
            const size_t o = ((size_t)i * MAX_PASSES + p) * 256;
            for (int d = 0; d < 256; ++d) { if (hh[o + d] == bt[i].n) trivial = true; hb[o + d] = run; run += hh[o + d]; }
            if (!trivial) all_trivial = false;
        }
        if (!all_trivial) todo.push_back(p);
    }
    int rc = HSK_OK;
    void *d_lookback = nullptr;
    u64 ntiles[XCD_BATCH];
    for (int i = 0; i < XCD_BATCH; ++i) ntiles[i] = bt[i].n < 2 ? 0 : (bt[i].n + TILE - 1) / TILE;
    std::vector<u32> tk((size_t)XCD_BATCH * MAX_PASSES + 64, 0);
    if (!todo.empty()) {
        HIPCHK(c, hipMemcpyAsync(d_gbase, hb.data(), hb.size() * 8, hipMemcpyHostToDevice, c->stream));
        const size_t lbw = wide ? 8 : 4;
        size_t lb_off[XCD_BATCH + 1]; lb_off[0] = 0;
        for (int i = 0; i < XCD_BATCH; ++i) lb_off[i + 1] = lb_off[i] + (size_t)ntiles[i] * 256 * lbw;
        const size_t per_pass = lb_off[XCD_BATCH];
        u64 max_tiles = 0; for (int i = 0; i < XCD_BATCH; ++i) max_tiles = std::max(max_tiles, ntiles[i]);
        const u32 grid = (u32)(XCD_BATCH * (max_tiles + max_tiles / 8) + 64);
        d_lookback = c->pool.alloc(per_pass * todo.size() + 256);
        if (!d_lookback) return fail(c, HSK_ERR_OOM, "look-back table of %zu bytes", per_pass * todo.size());
        HIPCHK(c, hipMemsetAsync(d_lookback, 0, per_pass * todo.size(), c->stream));
        u64 *kin[XCD_BATCH], *kout[XCD_BATCH], *vin[XCD_BATCH], *vout[XCD_BATCH];
        for (int i = 0; i < XCD_BATCH; ++i) { kin[i] = bt[i].kA; kout[i] = bt[i].kB; vin[i] = bt[i].vA; vout[i] = bt[i].vB; }
        for (size_t j = 0; j < todo.size(); ++j) {
            const int p = todo[j];
            MultiSortArgs m; memset(&m, 0, sizeof m);
            for (int i = 0; i < XCD_BATCH; ++i) {
                SortArgs &a = m.t[i];
                a.keys_in = kin[i]; a.keys_out = kout[i]; a.vals_in = vin[i]; a.vals_out = vout[i]; a.n = bt[i].n; a.ntiles = ntiles[i];
                a.word = plan[p].word; a.shift = plan[p].shift; a.bits = plan[p].bits;
                a.unstable = (hybrid && finish_follows && j == 0 && todo.size() > 1 && unstable_first_pass()) ? 1 : 0;
                a.gbase = d_gbase + ((size_t)i * MAX_PASSES + p) * 256;
                a.lookback = (char *)d_lookback + j * per_pass + lb_off[i];
                a.ticket = d_tickets + (size_t)i * MAX_PASSES + j; a.err = c->d_err;
            }
            EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 0; ep.keys = ntot; ep.bytes = 2 * ntot * (NW * 8 + (has_val ? 8 : 0)); (void)hipEventRecord(ep.a, c->stream); }
            if (has_val) { if (wide) launch_onesweep_multi<NW, true, u64>(c, m, grid); else launch_onesweep_multi<NW, true, u32>(c, m, grid); }
            else { if (wide) launch_onesweep_multi<NW, false, u64>(c, m, grid); else launch_onesweep_multi<NW, false, u32>(c, m, grid); }
            if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
            for (int i = 0; i < XCD_BATCH; ++i) { if (ntiles[i]) { std::swap(kin[i], kout[i]); std::swap(vin[i], vout[i]); } }
        }
        HIPCHK(c, hipGetLastError());
        for (int i = 0; i < XCD_BATCH; ++i) { bt[i].out_k = kin[i]; bt[i].out_v = vin[i]; }
    }
    u32 *d_flags = d_tickets + (size_t)XCD_BATCH * MAX_PASSES;          // [8] mixed-giant flags (zeroed with the tickets)
    if (hybrid && !finish_follows) {
        for (int i = 0; i < XCD_BATCH; ++i) {
            if (bt[i].n < 2) continue;
            u64 *other = (bt[i].out_k == bt[i].kA) ? bt[i].kB : bt[i].kA;
            u64 *vother = has_val ? ((bt[i].out_v == bt[i].vA) ? bt[i].vB : bt[i].vA) : nullptr;
            BinSortArgs b; b.in = bt[i].out_k; b.out = other; b.vin = bt[i].out_v; b.vout = vother; b.n = bt[i].n; b.hi_shift = HYBRID_SHIFT; b.mixed_giant = d_flags + i;
            hipLaunchKernelGGL(binsort_kernel, dim3((u32)((bt[i].n + BS_TILE - 1) / BS_TILE)), dim3(BS_THREADS), 0, c->stream, b);
            bt[i].out_k = other; bt[i].out_v = vother;
        }
        HIPCHK(c, hipGetLastError());
    }
    if (!todo.empty() || hybrid) {
        // every XCD must have drained its task (ticket counters >= tile counts); hybrid: which tasks need the long way
        HIPCHK(c, hipMemcpyAsync(tk.data(), d_tickets, ((size_t)XCD_BATCH * MAX_PASSES + XCD_BATCH) * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hsk_sync(c, c->stream));
        for (int i = 0; i < XCD_BATCH && rc == HSK_OK; ++i)
            for (size_t j = 0; j < todo.size(); ++j)
                if (tk[(size_t)i * MAX_PASSES + j] < ntiles[i]) { rc = fail(c, HSK_ERR_INTERNAL, "XCD %d did not drain its sort task (pass %zu: %u of %llu tiles)", i, j, tk[(size_t)i * MAX_PASSES + j], (unsigned long long)ntiles[i]); break; }
        if (hybrid && !finish_follows && rc == HSK_OK) {
            for (int i = 0; i < XCD_BATCH && rc == HSK_OK; ++i) {
                if (!tk[(size_t)XCD_BATCH * MAX_PASSES + i]) continue;
                c->stats.redone_tasks++;
                SortScratch sc1; rc = alloc_sort_scratch(c, sc1); if (rc) break;
                u64 *cur = bt[i].out_k, *other = (cur == bt[i].kA) ? bt[i].kB : bt[i].kA, *sk, *sv;
                u64 *vcur = bt[i].out_v, *vother = has_val ? ((vcur == bt[i].vA) ? bt[i].vB : bt[i].vA) : nullptr;
                rc = sort_task_device<NW>(c, cur, other, vcur, vother, bt[i].n, K, sc1, &sk, &sv, false);
                bt[i].out_k = sk; bt[i].out_v = sv;
                free_sort_scratch(c, sc1);
            }
        }
    }
    c->pool.release(d_lookback); if (!d_ghist_pre) c->pool.release(d_ghist); c->pool.release(d_gbase); c->pool.release(d_tickets);
    return rc;
}

static int alloc_sort_scratch(hsk_ctx *c, SortScratch &sc)
{
    DALLOC(c, sc.ghist, u64 *, (size_t)MAX_PASSES * 256 * 8);
    DALLOC(c, sc.gbase, u64 *, (size_t)MAX_PASSES * 256 * 8);
    DALLOC(c, sc.tickets, u32 *, 256);
    return HSK_OK;
}
static void free_sort_scratch(hsk_ctx *c, SortScratch &sc)
{
    c->pool.release(sc.ghist); c->pool.release(sc.gbase); c->pool.release(sc.tickets); c->pool.release(sc.lookback);
    sc = SortScratch();
}

static int check_device_error(hsk_ctx *c)
{
    u32 *e = (u32 *)((char *)c->pinned + c->pinned_bytes - 64);
    HIPCHK(c, hipMemcpyAsync(e, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hsk_sync(c, c->stream));
    if (*e) {
        (void)hipMemsetAsync(c->d_err, 0, 4, c->stream);
        const u32 w = *e;
        return fail(c, HSK_ERR_INTERNAL, "device-side check failed (error word %u:%s%s%s%s%s)", w, (w & 1) ? " radix look-back timed out;" : "",
                    (w & 2) ? " chunk map wait timed out;" : "", (w & 4) ? " foreign supermer;" : "",
                    (w & 8) ? " an XCD did not expand its task (cursors / histogram do not add up);" : "", (w & 16) ? " an XCD did not drain its sort task;" : "");
    }
    return HSK_OK;
}
