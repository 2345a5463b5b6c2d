// hsk_host_ctx.h -- host side: device memory pool, context, error plumbing, event pool, phase timer.
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
// (the device memory pool: hsk_pool.h -- segments of the regions hipMalloc returned, best fit with split and coalesce)
static int pool_hip_malloc(void **p, size_t n) { if (hipMalloc(p, n) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return 1; } return 0; }
static int pool_hip_free(void *p) { return hipFree(p) == hipSuccess ? 0 : 1; }

// pinned host memory for results (hipHostMalloc of gigabytes takes longer than counting them: freed result blocks are kept)
struct HostPool {
    std::multimap<size_t, void *> free_blocks;
    std::map<void *, size_t> live;
    void *alloc(size_t bytes)
    {
        if (bytes == 0) bytes = 64;
        // sizes are rounded up to 1/8 of their power of two: the result of the next call (a few per cent larger or
        // smaller) lands in the same size class and takes this call's block instead of pinning gigabytes again (~0.2 s per GB)
        { size_t g = 4096; while (g * 16 <= bytes) g <<= 1; bytes = (bytes + g - 1) & ~(g - 1); }
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 2 + (1u << 20)) {
            void *p = it->second; live[p] = it->first; bytes_cached -= it->first; free_blocks.erase(it); return p;
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            trim();
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        live[p] = bytes;
        return p;
    }
    // cached (free) pinned blocks are capped: a 10 Gbp call leaves ~8 GB of result and staging blocks behind, which the next call of
    // the same size takes again; beyond the cap the largest cached blocks go back to the system
    static constexpr size_t CACHE_CAP = (size_t)24 << 30;
    size_t bytes_cached = 0;
    void release(void *p)
    {
        auto it = live.find(p);
        if (it == live.end()) return;
        free_blocks.insert({it->second, p}); bytes_cached += it->second; live.erase(it);
        while (bytes_cached > CACHE_CAP && !free_blocks.empty()) { auto big = std::prev(free_blocks.end()); (void)hipHostFree(big->second); bytes_cached -= big->first; free_blocks.erase(big); }
    }
    void trim() { for (auto &kv : free_blocks) (void)hipHostFree(kv.second); free_blocks.clear(); bytes_cached = 0; }
    // hsk_destroy: the cache goes; blocks of results the caller still holds are NOT freed here (they belong to the results:
    // hsk_result_free(NULL, &res) releases them after the context is gone, include/hsk.h)
    void destroy() { trim(); live.clear(); }
};

struct EvPair { hipEvent_t a, b; int kind; u64 keys; u64 bytes; };

// What a call learns about its input BEFORE it commits to a plan (estimate_plan, hsk_api.hip): the k-mer spectrum of a small prefix of
// the reads, counted by the instance path, extrapolated to the whole input -- distinct k-mers per k-mer instance is what the combining
// extraction (pairs per k-mer), the first table of the LDS aggregation (distinct keys per prefix bin) and the decision to aggregate at
// all depend on.  Valid for the call that made it; the adaptive fields of the context are only the fallback where no estimate is made.
struct PlanEstimate {
    bool valid = false;
    double distinct_per_kmer = 0;      // estimated (distinct canonical k-mers) / (k-mer instances) of the WHOLE input
    double lambda_sample = 0;          // mean copies per genomic k-mer inside the sample
    double fraction = 0;               // sample bytes / input bytes
    u64 sample_kmers = 0, n1 = 0, n2 = 0, n3 = 0, distinct_sample = 0;
    u64 homo_at = 0, homo_cg = 0;      // exact copies of the all-A / all-C k-mer inside the sample (valid even where `valid` is not: payloads, wide keys)
    double ms = 0;                     // host wall clock of the estimate
};

// Tuning: "name=value,name=value" from hsk_config::tuning (copied at hsk_init), behind it the environment's HSK_TUNING (ad-hoc diagnostics
// without touching the client).  Forced paths for the byte-identity tests and a handful of measured thresholds; never an algorithm a caller
// would choose (those are hsk_config fields and HSK_FLAG_*).  Read per CONTEXT: two contexts of one process may differ (round 3 read ~35
// environment variables once per process into function-local statics).  INTEGRATION.md section 5 lists the names.
struct Tuning {
    std::map<std::string, long long> v;
    void parse(const char *s)
    {
        while (s && *s) {
            const char *e = strchr(s, ','); const std::string item = e ? std::string(s, e - s) : std::string(s);
            const size_t q = item.find('=');
            if (q != std::string::npos && q > 0) { std::string k = item.substr(0, q); while (!k.empty() && k[0] == ' ') k.erase(0, 1); if (!v.count(k)) v[k] = atoll(item.c_str() + q + 1); }
            s = e ? e + 1 : nullptr;
        }
    }
    long long get(const char *name, long long dflt) const { auto it = v.find(name); return it == v.end() ? dflt : it->second; }
};

struct hsk_ctx {
    hsk_config cfg;
    Tuning tune;
    int nw = 1;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;
    hipStream_t d2h_stream = nullptr;  // result copies, overlapped with the kernels of the following batches
    DevPool pool;
    HostPool hpool;
    double entries_per_kmer = 0;       // kept entries per input k-mer of the last hsk_count (sizes the pinned result block of the next one)
    char err[512] = {0};
    hsk_stats stats;
    std::vector<hipEvent_t> ev_free;
    std::vector<EvPair> ev_pending;
    void *pinned = nullptr; size_t pinned_bytes = 0;     // small staging area (histograms, totals)
    u32 *d_err = nullptr;
    Comm comm;
    const u8 *zc_src = nullptr;        // hsk_count() with pinned input: device view of the caller's packed reads (scan_kernel reads them in place)
    const u8 *h2d_src = nullptr;       // hsk_count() with pinned input, slab ingest: the caller's packed reads (host pointer); parse_count copies them slab by
    int h2d_slabs = 0;
    u32 scan_blocks = 0;               // experiments (hsk_debug_parse_overlap): workgroups of the parse kernels instead of 1024                 // slab (DMA, d2h_stream) and hashes slab s while slab s + 2 is on the link
    // hsk_count() with derived read offsets: host threads compare the caller's offsets with the back-to-back layout while the GPU
    // scans (result collected with the task totals); roff_host / roff_given: the caller's array and a device buffer for it, used
    // only when the comparison fails (a buffer with gaps)
    std::future<bool> roff_check;
    bool roff_bad = false;
    const uint32_t *rlen_host = nullptr;   // the read lengths were generated on the device from a sample (all reads equally long): the caller's array,
                                           // copied after all if the host threads find a read of another length (same fallback as a buffer with gaps)             // ... its verdict, when somebody other than parse_count collected it (parse_ingest_pipelined)
    const uint64_t *roff_host = nullptr;
    u64 *roff_given = nullptr;
    bool index_unchecked = false;      // hsk_count(): the read index is validated on the device (index_check_kernel), the verdict is read with the task totals
    int agg_first_cap = 10;            // log2 of the hash table the next aggregation starts with (AG_LOG2CAP_*): follows the fullest bin of the
                                       // previous batch, so that reads with errors / low coverage do not pay for a table they overflow anyway
    bool xcd_batch_ok = true;          // hsk_init's census saw workgroups on all eight XCC ids (see xcc_census_kernel)
    int agg_clean_batches = 0;         // EXTENSION: batches in a row whose first table held every bin (four of them: one table size down again)
    int agg_off_calls = 0;             // calls since agg_off / agg_off_wide was set
    bool agg_off_wide = false;         // the same for multi-word keys and EXTENSION: no prefix passes + tables, the full-width passes and the two-pass counter
    bool agg_off = false;              // one-word keys without payload: the input has too few copies per k-mer for the LDS aggregation (most bins of a
                                       // batch overflowed the 2048-slot table): batches take four prefix passes + the tile finish from here on
    // combining extraction (hsk_combine.h): combine_now = this call lays the store out for it (one GPU, one-word keys, no payload);
    // combine_off = the input kept too many pairs per k-mer (or a bin beat the weighted finish): the instance path until another look
    bool combine_now = false, combine_off = false; int combine_off_calls = 0;
    // ... another look after combine_off_period calls: 8, and twice as many every time the look finds the same kind of input again (up
    // to 64: a look costs the call ~2 x, reads with errors should not pay that every eighth call); back to 8 once a call has gone through
    int combine_off_period = 8, combine_good_calls = 0;
    bool combine_left_now = false;     // ... during THIS call (binding for the attempts that follow, whatever the estimate said)
    u32 drop_mask_now = 0;             // this call: bit 0 / 1 = the all-A / all-C k-mer has more than U copies inside the sample alone, the scan leaves its instances out
    u64 dropped_now = 0;               // ... and how many it left out (they count as k-mers of the input: hsk_result::total_kmers)
    void leave_combine() { combine_left_now = true; combine_off_period = combine_good_calls ? 8 : std::min(combine_off_period * 2, 64); combine_good_calls = 0; combine_off = true; combine_off_calls = 0; }
    bool item_mode_now = false;        // ... on ONE GPU: the store holds items (several ranks: byte runs + minimizer bits, items built by the owners)
    u32 vt_shift = 0;                  // this call's parse splits every task into 1 << vt_shift virtual tasks (combining extraction)
    int combine_prefix_floor = 0;      // ... never below this again (set when a bin beat the last table with fewer bits)
    int combine_prefix = 0;            // key bits of the weighted finish's bins the next batch is planned with (0: the default; follows the pairs per task)
    bool combine_veto = false;         // this call's store turned out to be no use to the combining extraction (too few tasks for a batch ...): the call again, without it
    bool pair_cap_full = false;        // this call's pair buffers ran over once: full size for the attempt that follows (dispatch_pipeline clears it with the call)
    PlanEstimate est;                  // this call's estimate (estimate_plan); est.valid decides instead of combine_off / agg_off / agg_first_cap
    double est_bias = 1.0;             // pairs per k-mer the combining extraction really produced / what the estimate promised, when a call had to leave the plan after all
    // a call that followed its estimate into the combining extraction and had to start again: estimates like this one are not believed again on this context
    void distrust_estimate(double ratio) { if (est.valid && est.distinct_per_kmer > 0) est_bias = std::max(est_bias, 1.05 / (est.distinct_per_kmer * ratio)); }
    int plan_attempt = 0;              // dispatch_pipeline: how often this call has been started again (HSK_RETRY_PLAN); bounded there
    bool forbid_long_way = false;      // heavy-hitter pre-aggregation: a task the aggregating finish cannot handle is reported, not redone
};

// HSK_FLAG_NO_AGGREGATION / HSK_FLAG_FULL_SORT of the context whose call is running on this thread (set by the counting entry
// points): the plan switches below (agg_enabled, hybrid_enabled, finish_enabled) are asked in places that have no context at hand
static thread_local int g_plan_flags = 0;
// ... and its tuning table (same lifetime: set by every entry point that takes a context)
static thread_local const Tuning *g_tune = nullptr;
static long long tune(const char *name, long long dflt) { return g_tune ? g_tune->get(name, dflt) : dflt; }
static void enter_ctx(hsk_ctx *c) { g_plan_flags = c->cfg.flags; g_tune = &c->tune; }

static int fail(hsk_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap; va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, HSK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define DALLOC(c, ptr, type, bytes)                                                              \
    do {                                                                                         \
        ptr = (type)(c)->pool.alloc(bytes);                                                      \
        if (!ptr) return fail(c, HSK_ERR_OOM, "device allocation of %zu bytes failed (%s:%d)", (size_t)(bytes), __FILE__, __LINE__); \
    } while (0)

// HSK_TIMING=1 (diagnostic): host-side wall-clock marks of one hsk_count call on stderr
#include <chrono>
constexpr int HSK_RETRY_PLAN = -2000;               // internal: the call is run again with another plan (dispatch_pipeline; never returned to the caller)
static bool timing_enabled() { static const bool on = getenv("HSK_TIMING") && atoi(getenv("HSK_TIMING")) != 0; return on; }
// a call that starts again with another plan says why when HSK_TIMING is set
static int retry_plan(const char *why, unsigned info = 0) { if (timing_enabled()) fprintf(stderr, "[hsk] the call starts again without the combining extraction: %s (%u)\n", why, info); return HSK_RETRY_PLAN; }
static void tmark(const char *what)
{
    if (!timing_enabled()) return;
    static auto t0 = std::chrono::steady_clock::now();
    static auto last = t0;
    const auto now = std::chrono::steady_clock::now();
    if (!what) { t0 = last = now; return; }
    fprintf(stderr, "[hsk %8.2f ms  +%7.2f] %s\n", std::chrono::duration<double, std::milli>(now - t0).count(), std::chrono::duration<double, std::milli>(now - last).count(), what);
    last = now;
}

// every blocking wait of the host on a stream inside the counting path goes through here (hsk_stats.host_syncs)
static hipError_t hsk_sync(hsk_ctx *c, hipStream_t s) { c->stats.host_syncs++; return hipStreamSynchronize(s); }

static hipEvent_t ev_get(hsk_ctx *c)
{
    if (!c->ev_free.empty()) { hipEvent_t e = c->ev_free.back(); c->ev_free.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
static void ev_put(hsk_ctx *c, hipEvent_t e) { c->ev_free.push_back(e); }

// events of one scope: handed back to the pool on every way out (the early returns of HIPCHK included)
struct EvList {
    hsk_ctx *c; std::vector<hipEvent_t> v;
    explicit EvList(hsk_ctx *c_) : c(c_) {}
    hipEvent_t get() { v.push_back(ev_get(c)); return v.back(); }
    ~EvList() { for (auto e : v) ev_put(c, e); }
};

// phase timer: records an event pair on the stream, elapsed time is summed after the final sync
struct PhaseTimer {
    hsk_ctx *c; std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs[8];
    explicit PhaseTimer(hsk_ctx *c_) : c(c_) {}
    void begin(int ph, hipStream_t s = nullptr) { hipEvent_t a = ev_get(c); (void)hipEventRecord(a, s ? s : c->stream); pairs[ph].push_back({a, nullptr}); }
    void end(int ph, hipStream_t s = nullptr) { hipEvent_t b = ev_get(c); (void)hipEventRecord(b, s ? s : c->stream); pairs[ph].back().second = b; }
    double collect(int ph)
    {
        double ms = 0;
        for (auto &p : pairs[ph]) {
            float f = 0;
            if (p.second && hipEventElapsedTime(&f, p.first, p.second) == hipSuccess) ms += f;
            ev_put(c, p.first); if (p.second) ev_put(c, p.second);
        }
        pairs[ph].clear();
        return ms;
    }
};
enum { PH_TOTAL = 0, PH_PARSE, PH_EXCH, PH_EXTRACT, PH_SORT, PH_COUNT, PH_D2H };
