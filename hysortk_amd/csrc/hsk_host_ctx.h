// hsk_host_ctx.h -- host side: device memory pool, context, error plumbing, event pool, phase timer.
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct DevPool {
    // freed blocks are kept and reused (hipMalloc/hipFree of multi-GB buffers costs milliseconds
    // and synchronises the device); exact-fit-or-slightly-larger reuse, trimmed on OOM/destroy.
    std::multimap<size_t, void *> free_blocks;
    std::map<void *, size_t> live;
    size_t bytes_live = 0, bytes_cached = 0, peak = 0;
    void *alloc(size_t bytes)
    {
        if (bytes == 0) bytes = 256;
        bytes = (bytes + 255) & ~(size_t)255;
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 4 + 4096) {
            void *p = it->second; size_t sz = it->first;
            free_blocks.erase(it); bytes_cached -= sz;
            live[p] = sz; bytes_live += sz; peak = std::max(peak, bytes_live);
            return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            trim();
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        live[p] = bytes; bytes_live += bytes; peak = std::max(peak, bytes_live);
        return p;
    }
    void release(void *p)
    {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;
        free_blocks.insert({it->second, p}); bytes_cached += it->second; bytes_live -= it->second;
        live.erase(it);
    }
    void trim()
    {
        for (auto &kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear(); bytes_cached = 0;
    }
    void destroy()
    {
        trim();
        for (auto &kv : live) (void)hipFree(kv.first);
        live.clear(); bytes_live = 0;
    }
};

struct EvPair { hipEvent_t a, b; int kind; u64 keys; u64 bytes; };

struct hsk_ctx {
    hsk_config cfg;
    int nw = 1;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;
    DevPool pool;
    char err[512] = {0};
    hsk_stats stats;
    std::vector<hipEvent_t> ev_free;
    std::vector<EvPair> ev_pending;
    void *pinned = nullptr; size_t pinned_bytes = 0;     // small staging area (histograms, totals)
    u32 *d_err = nullptr;
    Comm comm;
    bool xcd_batch_ok = true;          // hsk_init's census saw workgroups on all eight XCC ids (see xcc_census_kernel)
    bool forbid_long_way = false;      // heavy-hitter pre-aggregation: a task the aggregating finish cannot handle is reported, not redone
};

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

static hipEvent_t ev_get(hsk_ctx *c)
{
    if (!c->ev_free.empty()) { hipEvent_t e = c->ev_free.back(); c->ev_free.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
static void ev_put(hsk_ctx *c, hipEvent_t e) { c->ev_free.push_back(e); }

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
