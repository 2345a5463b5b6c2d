// hsk_api.hip -- host orchestration + C ABI of libhsk.so (see include/hsk.h).
//
// One translation unit, compiled with `hipcc --offload-arch=gfx950`.  The pipeline behind
// hsk_count() / hsk_count_device():
//
//   parse COUNT -> scan -> parse EMIT          (hsk_parse.h)   reads -> per-task supermers
//   [RCCL all-to-all of supermers]             (hsk_comm.h)    multi-GPU only
//   per owned task, ascending id:
//     expand (tile sums, scan, extract)        (hsk_expand.h)  supermers -> canonical k-mers
//     hist + onesweep passes                   (hsk_sort.h)    LSD radix sort
//     count COUNT -> scan -> EMIT              (hsk_count.h)   merge-count + [L,U] filter
//   result assembly (pinned host memory)
//
// There is no CPU fallback anywhere in this file: without a gfx950 device hsk_init() fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/hsk.h"
#include "hsk_device.h"
#include "hsk_parse.h"
#include "hsk_expand.h"
#include "hsk_sort.h"
#include "hsk_count.h"
#include "hsk_finish.h"
#include "hsk_agg.h"
#include "hsk_heavy.h"
#include "hsk_synth.h"
#include "hsk_plan.h"
#include "hsk_comm.h"

using namespace hsk;

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

// ------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" int hsk_abi_version(void) { return HSK_ABI_VERSION; }

extern "C" const char *hsk_strerror(int s)
{
    switch (s) {
    case HSK_OK: return "ok";
    case HSK_ERR_INVALID_ARG: return "invalid argument";
    case HSK_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
    case HSK_ERR_HIP: return "HIP runtime error";
    case HSK_ERR_OOM: return "out of memory";
    case HSK_ERR_DISPATCH: return "Cannot dispatch tasks. May be too unbalanced.";
    case HSK_ERR_INTERNAL: return "internal device-side check failed";
    case HSK_ERR_COMM: return "RCCL communication error";
    case HSK_ERR_UNSUPPORTED: return "not supported";
    default: return "unknown status";
    }
}

extern "C" const char *hsk_last_error(const hsk_ctx *c) { return c ? c->err : ""; }

extern "C" void hsk_config_default(hsk_config *cfg)
{
    memset(cfg, 0, sizeof *cfg);
    cfg->kmer_size = 31; cfg->minimizer_size = 17; cfg->lower_freq = 15; cfg->upper_freq = 40;   // reference Makefile:1-8
    cfg->extension = 0; cfg->ntasks = 0; cfg->device = 0; cfg->plain_dispatcher = 0;
    cfg->dispatch_upper_coe = 1.5; cfg->dispatch_step = 0.05; cfg->radix_bits = 8; cfg->flags = 0;
}

static int validate_cfg(const hsk_config *cfg)
{
    const int K = cfg->kmer_size, M = cfg->minimizer_size;
    if (!(2 < K && K < 96) || (K % 32) == 0) return 0;            // compiletime.h:10; K%32==0 is UB in the reference
    if (!(0 < M && M < K) || M > 31) return 0;                      // Makefile:50-52
    if (!(0 < cfg->lower_freq && cfg->lower_freq <= cfg->upper_freq && cfg->upper_freq <= 65535)) return 0;  // compiletime.h:21
    if (cfg->extension != 0 && cfg->extension != 1) return 0;
    if (cfg->ntasks < 0 || cfg->ntasks > HSK_MAX_TASKS) return 0;
    if (cfg->radix_bits != 0 && (cfg->radix_bits < 4 || cfg->radix_bits > 8)) return 0;
    return 1;
}

extern "C" int hsk_init(const hsk_config *cfg, hsk_ctx **out)
{
    if (!cfg || !out) return HSK_ERR_INVALID_ARG;
    *out = nullptr;
    if (!validate_cfg(cfg)) return HSK_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return HSK_ERR_NO_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) return HSK_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return HSK_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return HSK_ERR_NO_DEVICE;    // kernels are built for gfx950 only
    hsk_ctx *c = new hsk_ctx();
    c->cfg = *cfg;
    if (c->cfg.radix_bits == 0) c->cfg.radix_bits = 8;
    if (c->cfg.dispatch_upper_coe <= 0) c->cfg.dispatch_upper_coe = 1.5;
    if (c->cfg.dispatch_step <= 0) c->cfg.dispatch_step = 0.05;
    c->nw = (cfg->kmer_size + 31) / 32;
    memset(&c->stats, 0, sizeof c->stats);
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return HSK_ERR_HIP; }
    c->pinned_bytes = 1 << 20;
    if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault) != hipSuccess) { delete c; return HSK_ERR_OOM; }
    c->d_err = (u32 *)c->pool.alloc(256);
    if (!c->d_err) { delete c; return HSK_ERR_OOM; }
    (void)hipMemsetAsync(c->d_err, 0, 256, c->stream);
    *out = c;
    return HSK_OK;
}

extern "C" void hsk_destroy(hsk_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();
    c->comm.destroy();
    for (auto e : c->ev_free) (void)hipEventDestroy(e);
    for (auto &p : c->ev_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    c->pool.destroy();
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    delete c;
}

static void drain_profile_events(hsk_ctx *c)
{
    for (auto &p : c->ev_pending) {
        float f = 0;
        if (hipEventElapsedTime(&f, p.a, p.b) == hipSuccess) {
            if (p.kind == 0) { c->stats.scatter_launches++; c->stats.scatter_keys += p.keys; c->stats.scatter_bytes += p.bytes; c->stats.scatter_ms += f; }
            else if (p.kind == 2) { c->stats.agg_launches++; c->stats.agg_bytes += p.bytes; c->stats.agg_ms += f; }
            else { c->stats.hist_launches++; c->stats.hist_bytes += p.bytes; c->stats.hist_ms += f; }
        }
        ev_put(c, p.a); ev_put(c, p.b);
    }
    c->ev_pending.clear();
}

extern "C" int hsk_get_stats(hsk_ctx *c, hsk_stats *out, int reset)
{
    if (!c || !out) return HSK_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(c->stream);
    drain_profile_events(c);
    *out = c->stats;
    if (reset) memset(&c->stats, 0, sizeof c->stats);
    return HSK_OK;
}

// ------------------------------------------------------------------------------------------------
// stage: parse (a4, a5, a6)
// ------------------------------------------------------------------------------------------------
struct SupermerStore {
    u32 ntasks = 0, nblocks = 0;
    u8 *sm_len = nullptr; u8 *sm_bytes = nullptr; u64 *sm_gpos = nullptr; u32 *sm_pos = nullptr; int32_t *sm_rid = nullptr;
    u64 tot_sup = 0, tot_bytes = 0, tot_kmers = 0;
    std::vector<u64> task_tot;    // [ntasks][3] supermers, bytes, kmers
    std::vector<u64> task_base;   // [ntasks][3] slot, byte, kmer bases (tasks stored in `order`)
    std::vector<u32> order;       // storage order of tasks (grouped by owner rank, ascending id)
};

static void free_store(hsk_ctx *c, SupermerStore &s)
{
    c->pool.release(s.sm_len); c->pool.release(s.sm_bytes); c->pool.release(s.sm_gpos); c->pool.release(s.sm_pos); c->pool.release(s.sm_rid);
    s.sm_len = s.sm_bytes = nullptr; s.sm_gpos = nullptr; s.sm_pos = nullptr; s.sm_rid = nullptr;
}

static ParseArgs make_parse_args(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen,
                                 u64 nreads, int64_t rid_base, u32 ntasks, u32 *nblocks_out)
{
    ParseArgs a; memset(&a, 0, sizeof a);
    a.packed = d_packed; a.packed_bytes = packed_bytes; a.roff = d_roff; a.rlen = d_rlen; a.nreads = nreads;
    a.k = c->cfg.kmer_size; a.m = c->cfg.minimizer_size; a.ntasks = ntasks; a.fm = make_fastmod(ntasks);
    a.ntiles = (packed_bytes * 4 + PARSE_TILE - 1) / PARSE_TILE;
    u32 nblocks = (u32)std::min<u64>(a.ntiles, 1024);
    a.tiles_per_block = (u32)((a.ntiles + nblocks - 1) / nblocks);
    nblocks = (u32)((a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block);
    a.rid_base = rid_base;
    *nblocks_out = nblocks;
    return a;
}

// The parse in two steps, so that a caller that needs the task sizes before it can fix the storage order
// (multi-GPU: sizes -> all-reduce -> dispatcher -> owner-grouped order) hashes the reads only once:
//   parse_count: minimizers + supermer boundaries of every tile; per-(workgroup, task) counts; task totals
//   parse_place: exclusive scan of the counts in the storage order `order`, supermers to their slots
// Fast path = scan_kernel + place_kernel (compact supermer records kept in between); the general path
// (M > 25, or a tile with more supermers than the record capacity) = parse_kernel<COUNT> + emit_kernel.
struct ParseJob {
    ParseArgs a; u32 nblocks = 0, ntasks = 0; bool fast = false, empty = true;
    const u64 *d_roff = nullptr; u64 nreads = 0; int64_t rid_base = 0;
    u64 *d_blk_cnt = nullptr; u16 *d_dest_cache = nullptr; u32 *d_tile_rec = nullptr, *d_tile_nrec = nullptr, *d_overflow = nullptr;
    u32 *d_tile_r0 = nullptr;     // EXTENSION: first read of every tile (hint for the (pos, rid) lookup)
    std::vector<u64> task_tot;    // [ntasks][3] supermers, bytes, kmers of this rank
};

static void parse_release(hsk_ctx *c, ParseJob &j)
{
    c->pool.release(j.d_blk_cnt); c->pool.release(j.d_dest_cache); c->pool.release(j.d_tile_rec); c->pool.release(j.d_tile_nrec); c->pool.release(j.d_overflow);
    c->pool.release(j.d_tile_r0); j.d_tile_r0 = nullptr;
    j.d_blk_cnt = nullptr; j.d_dest_cache = nullptr; j.d_tile_rec = j.d_tile_nrec = j.d_overflow = nullptr;
}

static bool parse_fast_enabled()
{
    static const bool on = !(getenv("HSK_PARSE_FAST") && atoi(getenv("HSK_PARSE_FAST")) == 0);
    return on;
}
static u32 parse_rec_cap()
{
    static const u32 cap = getenv("HSK_PARSE_REC_CAP") ? (u32)std::min(std::max(atoi(getenv("HSK_PARSE_REC_CAP")), 1), (int)PLACE_MAX_REC) : SCAN_REC_CAP;
    return cap;
}

static int parse_count(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                       int64_t rid_base, u32 ntasks, ParseJob &j)
{
    j = ParseJob();
    j.ntasks = ntasks; j.d_roff = d_roff; j.nreads = nreads; j.rid_base = rid_base;
    j.task_tot.assign((size_t)ntasks * 3, 0);
    if (nreads == 0 || packed_bytes == 0) return HSK_OK;            // nothing to parse on this rank
    j.empty = false;
    j.a = make_parse_args(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, ntasks, &j.nblocks);
    ParseArgs &a = j.a;
    DALLOC(c, j.d_blk_cnt, u64 *, (size_t)j.nblocks * ntasks * 3 * 8);
    a.blk_cnt = j.d_blk_cnt;
    u64 *d_task_tot; DALLOC(c, d_task_tot, u64 *, (size_t)ntasks * 3 * 8 + 64);
    j.fast = parse_fast_enabled() && c->cfg.minimizer_size <= SCAN_MAX_M;
    u32 *h_ovf = (u32 *)((char *)c->pinned + c->pinned_bytes - 320);
    *h_ovf = 0;
    if (j.fast) {
        a.rec_cap = parse_rec_cap();
        a.place_group = std::max<u32>(1, std::min<u32>(16, PLACE_MAX_REC / a.rec_cap));
        DALLOC(c, j.d_tile_rec, u32 *, (size_t)a.ntiles * a.rec_cap * 4 + 64);
        DALLOC(c, j.d_tile_nrec, u32 *, (size_t)a.ntiles * 4 + 64);
        DALLOC(c, j.d_overflow, u32 *, 256);
        HIPCHK(c, hipMemsetAsync(j.d_overflow, 0, 4, c->stream));
        a.tile_rec = j.d_tile_rec; a.tile_nrec = j.d_tile_nrec; a.overflow = j.d_overflow;
        if (c->cfg.extension && nreads < (1ULL << 32)) { DALLOC(c, j.d_tile_r0, u32 *, (size_t)a.ntiles * 4 + 64); a.tile_r0 = j.d_tile_r0; }
        hipLaunchKernelGGL(scan_kernel, dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        hipLaunchKernelGGL(task_totals_kernel, dim3(1), dim3(HSK_MAX_TASKS), 0, c->stream, j.d_blk_cnt, j.nblocks, ntasks, d_task_tot);
        HIPCHK(c, hipMemcpyAsync(h_ovf, j.d_overflow, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(j.task_tot.data(), d_task_tot, (size_t)ntasks * 3 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*h_ovf) {                                                // a tile with more supermers than the record capacity
            j.fast = false; c->stats.parse_fallbacks++;
            c->pool.release(j.d_tile_r0); j.d_tile_r0 = nullptr; a.tile_r0 = nullptr;
            c->pool.release(j.d_tile_rec); c->pool.release(j.d_tile_nrec); j.d_tile_rec = j.d_tile_nrec = nullptr;
            a.tile_rec = a.tile_nrec = nullptr;
        }
    }
    if (!j.fast) {
        // task id per base position, kept from COUNT to EMIT (2 B x 4 x packed_bytes); optional: without it EMIT re-hashes
        j.d_dest_cache = (u16 *)c->pool.alloc((size_t)a.ntiles * PARSE_TILE * 2);
        a.dest_cache = j.d_dest_cache;
        hipLaunchKernelGGL((parse_kernel<PARSE_COUNT, false>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        hipLaunchKernelGGL(task_totals_kernel, dim3(1), dim3(HSK_MAX_TASKS), 0, c->stream, j.d_blk_cnt, j.nblocks, ntasks, d_task_tot);
        HIPCHK(c, hipMemcpyAsync(j.task_tot.data(), d_task_tot, (size_t)ntasks * 3 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipGetLastError());
    c->pool.release(d_task_tot);
    return HSK_OK;
}

// `order` (storage order of tasks) must be a permutation of 0..ntasks-1.
// skip (optional, [ntasks]): tasks whose supermers are not stored (they take no room and report zero totals)
static int parse_place(hsk_ctx *c, ParseJob &j, const std::vector<u32> &order, SupermerStore &st, const std::vector<u8> *skip = nullptr)
{
    const bool ext = c->cfg.extension != 0;
    const u32 ntasks = j.ntasks;
    st = SupermerStore();
    st.ntasks = ntasks; st.nblocks = j.nblocks; st.order = order;
    st.task_tot = j.task_tot;
    if (skip) for (u32 t = 0; t < ntasks; ++t) if ((*skip)[t]) st.task_tot[3 * t] = st.task_tot[3 * t + 1] = st.task_tot[3 * t + 2] = 0;
    st.task_base.assign((size_t)ntasks * 3, 0);
    { u64 s = 0, b = 0, k = 0;
      for (u32 i = 0; i < ntasks; ++i) { const u32 t = order[i]; st.task_base[3 * t] = s; st.task_base[3 * t + 1] = b; st.task_base[3 * t + 2] = k;
                                          s += st.task_tot[3 * t]; b += st.task_tot[3 * t + 1]; k += st.task_tot[3 * t + 2]; }
      st.tot_sup = s; st.tot_bytes = b; st.tot_kmers = k; }
    if (j.empty) return HSK_OK;
    ParseArgs &a = j.a;
    u64 *d_blk_base, *d_task_tot, *d_task_base; u32 *d_order;
    DALLOC(c, d_blk_base, u64 *, (size_t)j.nblocks * ntasks * 2 * 8);
    DALLOC(c, d_task_tot, u64 *, (size_t)ntasks * 3 * 8);
    DALLOC(c, d_task_base, u64 *, (size_t)ntasks * 3 * 8);
    DALLOC(c, d_order, u32 *, (size_t)ntasks * 4);
    HIPCHK(c, hipMemcpyAsync(d_order, order.data(), (size_t)ntasks * 4, hipMemcpyHostToDevice, c->stream));
    u8 *d_skip = nullptr;
    if (skip) {
        DALLOC(c, d_skip, u8 *, ntasks);
        HIPCHK(c, hipMemcpyAsync(d_skip, skip->data(), ntasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));          // the mask is host memory of the caller
    }
    a.task_skip = d_skip;
    hipLaunchKernelGGL(parse_scan_kernel, dim3(1), dim3(HSK_MAX_TASKS), 0, c->stream, j.d_blk_cnt, j.nblocks, ntasks, d_order, d_skip, d_task_tot, d_task_base, d_blk_base);
    DALLOC(c, st.sm_len, u8 *, st.tot_sup + 64);
    DALLOC(c, st.sm_gpos, u64 *, st.tot_sup * 8 + 64);          // reference mode: bases stay in the packed reads
    if (ext) { DALLOC(c, st.sm_pos, u32 *, st.tot_sup * 4 + 64); DALLOC(c, st.sm_rid, int32_t *, st.tot_sup * 4 + 64); }
    a.blk_base = d_blk_base; a.sm_len = st.sm_len; a.sm_gpos = st.sm_gpos;
    if (st.tot_sup) {
        if (j.fast) hipLaunchKernelGGL(place_kernel, dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16 + PLACE_MAX_REC * 4, c->stream, a);
        else if (a.dest_cache) hipLaunchKernelGGL(emit_kernel, dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        else hipLaunchKernelGGL((parse_kernel<PARSE_EMIT, false>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        if (ext) hipLaunchKernelGGL(resolve_pos_rid_kernel, dim3((u32)std::min<u64>((st.tot_sup + 255) / 256, 8192)), dim3(256), 0, c->stream,
                                    st.sm_gpos, st.tot_sup, j.d_roff, j.nreads, j.rid_base, st.sm_pos, st.sm_rid, (const u32 *)j.d_tile_r0, a.ntiles);
    }
    HIPCHK(c, hipGetLastError());
    // the small matrices are released after the stream has consumed them (pool reuse is stream-ordered:
    // every later user of these blocks is enqueued on the same stream)
    c->pool.release(d_blk_base); c->pool.release(d_task_tot); c->pool.release(d_task_base); c->pool.release(d_order); c->pool.release(d_skip);
    a.task_skip = nullptr;
    return HSK_OK;
}

// count + place with a known storage order
static int parse_phase(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                       int64_t rid_base, u32 ntasks, const std::vector<u32> &order, SupermerStore &st)
{
    ParseJob j;
    int rc = parse_count(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, ntasks, j);
    if (rc == HSK_OK) rc = parse_place(c, j, order, st);
    parse_release(c, j);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// stage: expand one task (a11)
// ------------------------------------------------------------------------------------------------
static void finalize_segs(TaskSegs &ts)
{
    u64 tile = 0;
    for (auto &s : ts.segs) { s.tile_start = tile; tile += (s.n_sup + EXP_TILE - 1) / EXP_TILE; }
    ts.ntiles = tile;
}

// where the bases of a task's supermers live
struct BaseSource {
    const u64 *src8 = nullptr; u64 bit0 = 0; u64 nwords = 0;   // byte stream rounded down to 8 bytes
    const u64 *gpos = nullptr;                                  // reference mode positions (null: prefix-sum byte offsets)
};
static BaseSource source_from_packed(const u8 *d_packed, u64 packed_bytes, const u64 *gpos)
{
    BaseSource b; const uintptr_t p = (uintptr_t)d_packed;
    b.src8 = (const u64 *)(p & ~(uintptr_t)7); b.bit0 = 8 * (u64)(p & 7); b.nwords = ((p & 7) + packed_bytes + 7) / 8; b.gpos = gpos;
    return b;
}
static BaseSource source_from_bytes(const u8 *bytes, u64 nbytes)
{
    BaseSource b; b.src8 = (const u64 *)bytes; b.bit0 = 0; b.nwords = (nbytes + 7) / 8 + 1; b.gpos = nullptr;   // pool blocks are padded
    return b;
}

struct ExpandScratch { ExpSeg *d_segs = nullptr; u64 *d_tile_sum = nullptr, *d_tile_off = nullptr; };

// tile sums + scans of n <= EXP_PREP_BATCH tasks with two launches
static int expand_prepare_batch(hsk_ctx *c, int n, const TaskSegs *const *ts, const u8 *const *sm_len, ExpandScratch *x,
                                hipStream_t stream = nullptr, bool prealloc = false)
{
    if (!stream) stream = c->stream;
    ExpandPrepArgs pa; memset(&pa, 0, sizeof pa);
    pa.k = c->cfg.kmer_size;
    u64 max_tiles = 0; int max_seg = 0;
    for (int i = 0; i < n; ++i) {
        const int nseg = (int)ts[i]->segs.size();
        if (!prealloc) {
            DALLOC(c, x[i].d_segs, ExpSeg *, sizeof(ExpSeg) * nseg);
            DALLOC(c, x[i].d_tile_sum, u64 *, ts[i]->ntiles * 16);
            DALLOC(c, x[i].d_tile_off, u64 *, ts[i]->ntiles * 16);
        }
        HIPCHK(c, hipMemcpyAsync(x[i].d_segs, ts[i]->segs.data(), sizeof(ExpSeg) * nseg, hipMemcpyHostToDevice, stream));
        pa.segs[i] = x[i].d_segs; pa.nseg[i] = nseg; pa.sm_len[i] = sm_len[i]; pa.ntiles[i] = ts[i]->ntiles;
        pa.tile_sum[i] = x[i].d_tile_sum; pa.tile_off[i] = x[i].d_tile_off;
        max_tiles = std::max(max_tiles, ts[i]->ntiles); max_seg = std::max(max_seg, nseg);
    }
    if (max_tiles == 0) return HSK_OK;
    hipLaunchKernelGGL(expand_tilesum_kernel, dim3((u32)max_tiles, n), dim3(EXP_THREADS), 0, stream, pa);
    hipLaunchKernelGGL(expand_scan_kernel, dim3(max_seg, n), dim3(EXP_THREADS), 0, stream, pa);
    return HSK_OK;
}
static int expand_prepare(hsk_ctx *c, const TaskSegs &ts, const u8 *sm_len, ExpandScratch &x)
{
    const TaskSegs *tp = &ts;
    return expand_prepare_batch(c, 1, &tp, &sm_len, &x);
}
static void expand_release(hsk_ctx *c, ExpandScratch &x) { c->pool.release(x.d_segs); c->pool.release(x.d_tile_sum); c->pool.release(x.d_tile_off); x = ExpandScratch(); }

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
    {
        const TaskSegs *tsp[EXP_BATCH]; const u8 *lens[EXP_BATCH]; int m = 0;
        for (int i = 0; i < njobs; ++i) if (jobs[i].ts->ntiles) { tsp[m] = jobs[i].ts; lens[m] = jobs[i].sm_len; ++m; }
        // (ts.segs is host memory owned by the caller and stays alive until the next sync)
        int rc = expand_prepare_batch(c, m, tsp, lens, x, stream, pre != nullptr); if (rc) return rc;
    }
    for (int i = 0; i < njobs; ++i) {
        const ExpandJob &j = jobs[i];
        if (j.ts->ntiles == 0) continue;
        ExpandTask &t = a.t[nt];
        t.segs = x[nt].d_segs; t.nseg = (int)j.ts->segs.size(); t.sm_len = j.sm_len;
        t.src8 = j.src.src8; t.src_bit0 = j.src.bit0; t.src_words = j.src.nwords; t.sm_gpos = j.src.gpos; t.sm_pos = j.sm_pos; t.sm_rid = j.sm_rid;
        t.tile_off = x[nt].d_tile_off; t.ntiles = j.ts->ntiles; t.keys_out = j.keys; t.vals_out = j.vals; t.ghist = npass ? j.ghist : nullptr;
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
    // pipelined with the sort of the previous batch (second stream): take only part of every CU, the rest is the sort's
    static const int share_pct = getenv("HSK_EXPAND_SHARE") ? atoi(getenv("HSK_EXPAND_SHARE")) : 100;
    const int occ_use = (stream != c->stream) ? std::max(1, occ * share_pct / 100) : occ;
    u32 rw = (u32)std::max(1, occ_use * 256 / (8 * nt));
    rw = (u32)std::min<u64>(rw, (max_tiles + 7) / 8);
    a.row_workers = std::max<u32>(rw, 1);
    a.nrows = max_tiles;
    const u32 grid = 8u * (u32)nt * a.row_workers;
    if (ext) hipLaunchKernelGGL((expand_kernel<NW, true>), dim3(grid), dim3(EXP_THREADS), dyn, stream, a);
    else hipLaunchKernelGGL((expand_kernel<NW, false>), dim3(grid), dim3(EXP_THREADS), dyn, stream, a);
    HIPCHK(c, hipGetLastError());
    if (!pre) for (int i = 0; i < nt; ++i) expand_release(c, x[i]);
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
static int pack_store_bytes(hsk_ctx *c, SupermerStore &st, const BaseSource &src)
{
    DALLOC(c, st.sm_bytes, u8 *, st.tot_bytes + 64);
    if (st.tot_sup == 0) return HSK_OK;
    TaskSegs all; ExpSeg s; s.sup_off = 0; s.n_sup = st.tot_sup; s.byte_off = 0; s.kmer_off = 0; s.tile_start = 0;
    all.segs.push_back(s);
    all.ntiles = (st.tot_sup + EXP_TILE - 1) / EXP_TILE;
    ExpandScratch x;
    int rc = expand_prepare(c, all, st.sm_len, x); if (rc) return rc;
    hipLaunchKernelGGL(pack_kernel, dim3((u32)all.ntiles), dim3(EXP_THREADS), 0, c->stream, x.d_segs, 1, st.sm_len, src.src8, src.bit0, src.nwords,
                       st.sm_gpos, x.d_tile_off, st.sm_bytes);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));          // `all` lives on this stack frame
    expand_release(c, x);
    return HSK_OK;
}

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
    static const bool on = !(getenv("HSK_HYBRID") && atoi(getenv("HSK_HYBRID")) == 0);
    return on;
}
// Two-word keys take the prefix plan when the aggregating finish follows and the most significant word carries at
// least the 16 prefix bits (K >= 40)
template <int NW> static bool prefix_plan_ok(int K, bool finish_follows)
{
    return hybrid_enabled() && (NW == 1 || (NW == 2 && finish_follows && K - 32 >= 8));
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
    static const int pad = getenv("HSK_SORT_LDS_PAD") ? atoi(getenv("HSK_SORT_LDS_PAD")) : 0;   // experiment knob: extra LDS per workgroup lowers residency
    hipLaunchKernelGGL((onesweep_kernel<NW, HAS_VAL, LB>), dim3(ntiles), dim3(SORT_THREADS), (size_t)pad, c->stream, a);
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    const bool wide = n >= (1ULL << 30);
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
        HIPCHK(c, hipStreamSynchronize(c->stream));
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
    u64 ntot = 0; bool wide = false;
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
        HIPCHK(c, hipStreamSynchronize(c->stream));
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

// ---- the one-pass plan: ONE scatter pass (top 8 bits) over `nb` tasks in ONE launch (onesweep_many_kernel) ------
// d_ghist: [nb][MAX_PASSES][256], histogram of pass 0 (bits 56..63) counted by expand_batch.  nb is a multiple of 8.
constexpr int MANY_MAX = 64;
template <int NW>
static int sort_many_onepass(hsk_ctx *c, BatchTask *bt, int nb, u64 *d_ghist)
{
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    constexpr int TILE = SortTile<NW>::TILE;
    const int per_xcd = nb / 8;
    std::vector<u64> hh((size_t)nb * MAX_PASSES * 256), hb((size_t)nb * 256, 0);
    HIPCHK(c, hipMemcpyAsync(hh.data(), d_ghist, hh.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    u64 ntiles[MANY_MAX]; size_t lb_off[MANY_MAX + 1]; lb_off[0] = 0;
    u64 ntot = 0;
    for (int i = 0; i < nb; ++i) {
        bt[i].out_k = bt[i].kA; bt[i].out_v = bt[i].vA;
        u64 run = 0; bool trivial = false;
        for (int d = 0; d < 256; ++d) { const u64 v = hh[((size_t)i * MAX_PASSES) * 256 + d]; if (v == bt[i].n) trivial = true; hb[(size_t)i * 256 + d] = run; run += v; }
        ntiles[i] = (bt[i].n < 2 || trivial) ? 0 : (bt[i].n + TILE - 1) / TILE;      // one digit value only: already "sorted"
        if (bt[i].n >= (1ULL << 30)) return fail(c, HSK_ERR_INTERNAL, "one-pass plan on a task of 2^30 keys");
        lb_off[i + 1] = lb_off[i] + (size_t)ntiles[i] * 256 * 4;
        if (ntiles[i]) ntot += bt[i].n;
    }
    if (lb_off[nb] == 0) return HSK_OK;
    u64 *d_gbase; u32 *d_tk, *d_pre; void *d_lb; SortArgs *d_tasks;
    DALLOC(c, d_gbase, u64 *, (size_t)nb * 256 * 8);
    DALLOC(c, d_tk, u32 *, (size_t)(nb + 8) * 4 + 64);                 // task tickets, then the 8 XCD counters
    DALLOC(c, d_pre, u32 *, (size_t)8 * (per_xcd + 1) * 4);
    DALLOC(c, d_lb, void *, lb_off[nb] + 256);
    DALLOC(c, d_tasks, SortArgs *, sizeof(SortArgs) * nb);
    HIPCHK(c, hipMemsetAsync(d_tk, 0, (size_t)(nb + 8) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(d_lb, 0, lb_off[nb], c->stream));
    HIPCHK(c, hipMemcpyAsync(d_gbase, hb.data(), hb.size() * 8, hipMemcpyHostToDevice, c->stream));
    std::vector<SortArgs> ta(nb); std::vector<u32> pre((size_t)8 * (per_xcd + 1), 0);
    u64 max_xcd = 0;
    for (int x = 0; x < 8; ++x) {
        u32 run = 0;
        for (int j = 0; j < per_xcd; ++j) { pre[(size_t)x * (per_xcd + 1) + j] = run; run += (u32)ntiles[x + 8 * j]; }
        pre[(size_t)x * (per_xcd + 1) + per_xcd] = run;
        max_xcd = std::max<u64>(max_xcd, run);
    }
    for (int i = 0; i < nb; ++i) {
        SortArgs &a = ta[i]; memset(&a, 0, sizeof a);
        a.keys_in = bt[i].kA; a.keys_out = bt[i].kB; a.vals_in = nullptr; a.vals_out = nullptr; a.n = bt[i].n; a.ntiles = ntiles[i];
        a.word = NW - 1; a.shift = 56; a.bits = 8;
        a.gbase = d_gbase + (size_t)i * 256; a.lookback = (char *)d_lb + lb_off[i]; a.ticket = d_tk + i; a.err = c->d_err;
    }
    HIPCHK(c, hipMemcpyAsync(d_tasks, ta.data(), sizeof(SortArgs) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_pre, pre.data(), pre.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                       // ta / pre / hb are host stack memory
    ManySortArgs m; m.tasks = d_tasks; m.xcd_prefix = d_pre; m.xcd_counter = d_tk + nb; m.per_xcd = per_xcd;
    const u32 grid = (u32)(8 * (max_xcd + max_xcd / 8) + 64);
    EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 0; ep.keys = ntot; ep.bytes = 2 * ntot * NW * 8; (void)hipEventRecord(ep.a, c->stream); }
    hipLaunchKernelGGL((onesweep_many_kernel<NW, false, u32>), dim3(grid), dim3(SORT_THREADS), 0, c->stream, m);
    if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
    HIPCHK(c, hipGetLastError());
    std::vector<u32> tk(nb + 8);
    HIPCHK(c, hipMemcpyAsync(tk.data(), d_tk, (size_t)(nb + 8) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = HSK_OK;
    for (int i = 0; i < nb && rc == HSK_OK; ++i) {
        if (tk[i] < ntiles[i]) rc = fail(c, HSK_ERR_INTERNAL, "XCD %d did not drain sort task %d (%u of %llu tiles)", i & 7, i, tk[i], (unsigned long long)ntiles[i]);
        if (ntiles[i]) bt[i].out_k = bt[i].kB;
    }
    c->pool.release(d_gbase); c->pool.release(d_tk); c->pool.release(d_pre); c->pool.release(d_lb); c->pool.release(d_tasks);
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (*e) {
        (void)hipMemsetAsync(c->d_err, 0, 4, c->stream);
        return fail(c, HSK_ERR_INTERNAL, "radix look-back timed out (device error word %u)", *e);
    }
    return HSK_OK;
}

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
    HIPCHK(c, hipStreamSynchronize(c->stream));
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

static void *host_alloc(ResultPriv *rp, size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    rp->host_blocks.push_back(p);
    return p;
}

static bool finish_enabled();
static bool agg_enabled();
// One scatter pass + aggregation over 8-bit prefix bins (hsk_agg.h: agg_big_kernel) for tasks of up to ONEPASS_MAX_TASK
// k-mers; HSK_ONEPASS=0 keeps two passes + 16-bit bins for every task.
constexpr u64 ONEPASS_TASK_KMERS = 1ULL << 24;          // auto_ntasks aims at this many base positions per task
constexpr u64 ONEPASS_MAX_TASK = 3ULL << 23;            // larger tasks: bins with too many distinct keys for the LDS table
// EXPERIMENTAL, off unless HSK_ONEPASS=1: at 10 Gbp it means ~600 tasks of 13 M k-mers, processed in batches of 64
// (one scatter launch per batch).  It removes 16 B of HBM traffic per k-mer and the scatter phase drops from 57 to 36
// ms, but the whole step is slower today (226 ms against 168 ms): the aggregation over 51 000-record bins runs one
// 1024-thread workgroup per CU (112 KB of LDS) and only reaches 20 % issue utilisation (82 ms against 27 ms), placing
// supermers into 600 tasks costs +8 ms and 76 small expand launches +8 ms.
static bool onepass_enabled()
{
    static const bool on = getenv("HSK_ONEPASS") && atoi(getenv("HSK_ONEPASS")) != 0;
    return on;
}

static u32 auto_ntasks(hsk_ctx *c, u64 packed_bytes, int nranks)
{
    // one task per ~2^28 k-mers (2 GB of 8-byte keys): large enough to saturate the chip, small
    // enough that key + ping-pong + look-back buffers of one task stay a small share of HBM
    u64 est = packed_bytes * 4 * (u64)std::max(nranks, 1);
    u64 t = (est + (1ULL << 28) - 1) >> 28;
    // one-word keys without payload: tasks small enough for ONE scatter pass + aggregation over 8-bit prefix bins
    // (as many as HSK_MAX_TASKS allows; beyond that the tasks grow and the two-pass plan takes over by itself)
    if (c->nw == 1 && c->cfg.extension == 0 && onepass_enabled() && hybrid_enabled() && finish_enabled() && agg_enabled())
        t = std::max(t, std::min<u64>((est + ONEPASS_TASK_KMERS - 1) / ONEPASS_TASK_KMERS, HSK_MAX_TASKS / 8 * 8));
    t = std::max<u64>(t, (u64)std::max(nranks, 1));
    // tasks are sorted eight at a time (one per XCD): give every rank a multiple of eight when there are that many
    const u64 per = 8ULL * (u64)std::max(nranks, 1);
    if (t >= per) t = (t + per - 1) / per * per;
    return (u32)std::min<u64>(std::max<u64>(t, 1), HSK_MAX_TASKS);
}

// Fused finish of a batch (hybrid sort, one-word keys, no payload): one finish_multi_kernel launch turns the
// prefix-ordered keys of eight tasks into their (k-mer, count) lists.  Tasks the kernel could not finish (a
// long bin with several keys, see hsk_finish.h) are redone with the full-width passes and the two-pass counter.
static bool finish_enabled()
{
    static const bool on = !(getenv("HSK_FUSED_FINISH") && atoi(getenv("HSK_FUSED_FINISH")) == 0);
    return on;
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    HIPCHK(c, hipStreamSynchronize(c->stream));         // scratch buffers are reused by the next batch
    for (int i = 0; i < XCD_BATCH; ++i) if (own_scratch[i]) c->pool.release(scratch[i]);
    c->pool.release(d_ctl);
    return rc;
}

// ---- two passes + aggregation (hsk_agg.h): the batch's keys are sorted on their top 16 bits ---------------
static bool agg_enabled()
{
    static const bool on = !(getenv("HSK_AGG") && atoi(getenv("HSK_AGG")) == 0);
    return on;
}

template <int NW>
// prefix_bits = 16: bins of the top 16 bits (two scatter passes), small tables with a retry ladder and the long way;
// prefix_bits = 8: bins of the top 8 bits (one scatter pass), agg_big_kernel; a task it cannot take is reported in
// outs[i].failed (the caller orders it on 8 more bits and comes back with prefix_bits = 16).
static int agg_finish_batch_device(hsk_ctx *c, BatchTask *bt, int K, u64 max_task, u64 *d_histo, u32 histo_len, TaskOut *outs, int prefix_bits = AG_PREFIX_BITS)
{
    static_assert(NW <= 2, "the aggregating finish handles one- and two-word keys");
    constexpr u32 EW = NW + 1;                          // words per entry
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    const u32 L = (u32)c->cfg.lower_freq;
    const u32 slot_shift = L >= 2 ? 1 : 0;              // a bin of n records keeps at most n / L entries of 16 bytes
    const bool big = prefix_bits == 8;
    const u32 nbins = 1u << prefix_bits;
    const size_t per = (size_t)nbins + 8;
    u64 *d_bounds, *d_cnt; u32 *d_flags;
    DALLOC(c, d_bounds, u64 *, per * 8 * AG_BATCH);
    DALLOC(c, d_cnt, u64 *, per * 8 * AG_BATCH);
    DALLOC(c, d_flags, u32 *, 256);
    HIPCHK(c, hipMemsetAsync(d_flags, 0, 64, c->stream));
    AggArgs a; memset(&a, 0, sizeof a);
    a.lower = L; a.upper = (u32)c->cfg.upper_freq; a.nbins = nbins; a.shift = 64 - prefix_bits; a.nw = NW;
    bool own_scratch[AG_BATCH] = {false};
    u64 ntot = 0;
    for (int i = 0; i < AG_BATCH; ++i) {
        AggTask &t = a.t[i];
        outs[i] = TaskOut();
        if (bt[i].n == 0) continue;
        u64 *other = (bt[i].out_k == bt[i].kA) ? bt[i].kB : bt[i].kA;
        t.keys = bt[i].out_k; t.n = bt[i].n; t.bounds = d_bounds + per * i; t.bin_cnt = d_cnt + per * i; t.flags = d_flags + i;
        t.slot_shift = slot_shift; t.active = 1; ntot += bt[i].n;
        if (slot_shift) t.scratch = other;               // the idle ping-pong buffer: n / 2 entries
        else {
            t.scratch = (u64 *)c->pool.alloc(bt[i].n * EW * 8 + 64); own_scratch[i] = true;
            if (!t.scratch) return fail(c, HSK_ERR_OOM, "finish scratch of %llu bytes", (unsigned long long)(bt[i].n * EW * 8));
        }
    }
    struct { u32 flags[AG_BATCH]; u64 total[AG_BATCH]; } h;
    auto run = [&](int log2cap) -> int {
        EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 2; ep.keys = ntot; ep.bytes = ntot * NW * 8; (void)hipEventRecord(ep.a, c->stream); }
        if (NW == 2) {
            if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_SMALL>), dim3(nbins, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
            else hipLaunchKernelGGL((agg2_finish_kernel<AG_LOG2CAP_LARGE>), dim3(nbins, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        } else
        if (big) hipLaunchKernelGGL(agg_big_kernel, dim3(nbins, AG_BATCH), dim3(AGB_THREADS), 0, c->stream, a);
        else if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_SMALL>), dim3(nbins, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((agg_finish_kernel<AG_LOG2CAP_LARGE>), dim3(nbins, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
        hipLaunchKernelGGL(agg_scan_kernel, dim3(AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h.flags, d_flags, sizeof h.flags, hipMemcpyDeviceToHost, c->stream));
        for (int i = 0; i < AG_BATCH; ++i) if (a.t[i].active) HIPCHK(c, hipMemcpyAsync(&h.total[i], a.t[i].bin_cnt + nbins, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return HSK_OK;
    };
    memset(&h, 0, sizeof h);
    hipLaunchKernelGGL(bin_bounds_kernel, dim3(nbins / AG_THREADS + 1, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    int rc = run(AG_LOG2CAP_SMALL); if (rc) return rc;
    bool retry = false, done[AG_BATCH];
    u64 total[AG_BATCH];
    for (int i = 0; i < AG_BATCH; ++i) { done[i] = bt[i].n == 0 || !h.flags[i]; total[i] = h.total[i]; if (!done[i]) retry = true; }
    if (retry && !big) {
        // second chance with the large table for the tasks that overflowed
        AggArgs keep = a;
        for (int i = 0; i < AG_BATCH; ++i) { a.t[i].active = (keep.t[i].active && !done[i]) ? 1 : 0; c->stats.agg_retried_tasks += a.t[i].active; }
        HIPCHK(c, hipMemsetAsync(d_flags, 0, 64, c->stream));
        memset(&h, 0, sizeof h);
        rc = run(AG_LOG2CAP_LARGE); if (rc) return rc;
        for (int i = 0; i < AG_BATCH; ++i) if (a.t[i].active) { done[i] = !h.flags[i]; total[i] = h.total[i]; }
        a = keep;
    }
    AggCompactArgs ca; memset(&ca, 0, sizeof ca);
    ca.slot_shift = slot_shift; ca.histo = d_histo; ca.histo_len = histo_len; ca.nbins = nbins; ca.ew = EW;
    bool any = false;
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || !done[i]) continue;
        c->stats.fused_tasks++;
        outs[i].n = total[i];
        if (outs[i].n) {
            outs[i].entries = (u64 *)c->pool.alloc(outs[i].n * EW * 8);
            if (!outs[i].entries) { rc = fail(c, HSK_ERR_OOM, "task output of %llu bytes", (unsigned long long)(outs[i].n * EW * 8)); break; }
            ca.scratch[i] = a.t[i].scratch; ca.bounds[i] = a.t[i].bounds; ca.bin_off[i] = a.t[i].bin_cnt; ca.entries[i] = outs[i].entries;
            any = true;
        }
    }
    if (any && rc == HSK_OK) hipLaunchKernelGGL(agg_compact_kernel, dim3(big ? 64 : 256, AG_BATCH), dim3(AG_THREADS), 0, c->stream, ca);
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || done[i]) continue;
        if (big || c->forbid_long_way) { outs[i].failed = true; continue; }
        // the long way for this task: full-width passes from the current order, then the two-pass counter
        c->stats.redone_tasks++;
        if (own_scratch[i]) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->pool.release(a.t[i].scratch); a.t[i].scratch = nullptr; own_scratch[i] = false; }
        SortScratch sc1; rc = alloc_sort_scratch(c, sc1); if (rc) break;
        u64 *cur = bt[i].out_k, *other = (cur == bt[i].kA) ? bt[i].kB : bt[i].kA, *sk, *sv;
        rc = sort_task_device<NW>(c, cur, other, nullptr, nullptr, bt[i].n, K, sc1, &sk, &sv, false);
        free_sort_scratch(c, sc1);
        if (rc == HSK_OK) rc = count_task_device<NW>(c, sk, nullptr, bt[i].n, 0, d_histo, histo_len, outs[i]);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));         // scratch buffers are reused by the next batch
    for (int i = 0; i < AG_BATCH; ++i) if (own_scratch[i]) c->pool.release(a.t[i].scratch);
    c->pool.release(d_bounds); c->pool.release(d_cnt); c->pool.release(d_flags);
    (void)max_task;
    return rc;
}

// ---- EXTENSION: two passes + grouping aggregation (hsk_agg.h: agg_ext_kernel) ---------------------------------------
// pay_before[i]: offset of task i's payload range in the rank's payload arrays (payload_off values are global over the
// owned tasks in ascending id).
static int agg_ext_finish_batch_device(hsk_ctx *c, BatchTask *bt, int K, const u64 *pay_before, u64 *d_histo, u32 histo_len, TaskOut *outs)
{
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
    a.lower = L; a.upper = (u32)c->cfg.upper_freq; a.nbins = nbins; a.shift = AG_SHIFT;
    sa.nbins = nbins; sa.shift = AG_SHIFT; sa.nw = 1;
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
        if (slot_shift) { t.scratch_e = other_k; t.scratch_p = other_v; }       // n / 2 entries of 16 + 8 bytes: the idle ping-pong buffers
        else {
            t.scratch_e = (u64 *)c->pool.alloc(bt[i].n * 16 + 64); t.scratch_p = (u64 *)c->pool.alloc(bt[i].n * 8 + 64); own_scratch[i] = true;
            if (!t.scratch_e || !t.scratch_p) return fail(c, HSK_ERR_OOM, "finish scratch");
        }
        outs[i].npay = bt[i].n;
        DALLOC(c, outs[i].pos, u32 *, bt[i].n * 4); DALLOC(c, outs[i].rid, int32_t *, bt[i].n * 4);
        t.pos = outs[i].pos; t.rid = outs[i].rid;
        sa.t[i].bin_cnt = t.bin_cnt; sa.t[i].active = 1;
    }
    struct { u32 flags[AG_BATCH]; u64 total[AG_BATCH]; } h;
    auto run = [&](int log2cap) -> int {
        for (int i = 0; i < AG_BATCH; ++i) sa.t[i].active = a.t[i].active;
        EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 2; ep.keys = ntot; ep.bytes = ntot * 16; (void)hipEventRecord(ep.a, c->stream); }
        if (log2cap == AG_LOG2CAP_SMALL) hipLaunchKernelGGL((agg_ext_kernel<AG_LOG2CAP_SMALL>), dim3(nbins, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((agg_ext_kernel<AG_LOG2CAP_LARGE>), dim3(nbins, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
        if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
        hipLaunchKernelGGL(agg_scan_kernel, dim3(AG_BATCH), dim3(AG_THREADS), 0, c->stream, sa);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h.flags, d_flags, sizeof h.flags, hipMemcpyDeviceToHost, c->stream));
        for (int i = 0; i < AG_BATCH; ++i) if (a.t[i].active) HIPCHK(c, hipMemcpyAsync(&h.total[i], a.t[i].bin_cnt + nbins, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return HSK_OK;
    };
    memset(&h, 0, sizeof h);
    hipLaunchKernelGGL(bin_bounds_ext_kernel, dim3(nbins / AG_THREADS + 1, AG_BATCH), dim3(AG_THREADS), 0, c->stream, a);
    int rc = run(AG_LOG2CAP_SMALL); if (rc) return rc;
    bool retry = false, done[AG_BATCH];
    u64 total[AG_BATCH];
    for (int i = 0; i < AG_BATCH; ++i) { done[i] = bt[i].n == 0 || !h.flags[i]; total[i] = h.total[i]; if (!done[i]) retry = true; }
    if (retry) {
        AggExtArgs keep = a;
        for (int i = 0; i < AG_BATCH; ++i) { a.t[i].active = (keep.t[i].active && !done[i]) ? 1 : 0; c->stats.agg_retried_tasks += a.t[i].active; }
        HIPCHK(c, hipMemsetAsync(d_flags, 0, 64, c->stream));
        memset(&h, 0, sizeof h);
        rc = run(AG_LOG2CAP_LARGE); if (rc) return rc;
        for (int i = 0; i < AG_BATCH; ++i) if (a.t[i].active) { done[i] = !h.flags[i]; total[i] = h.total[i]; }
        a = keep;
    }
    AggExtCompactArgs ca; memset(&ca, 0, sizeof ca);
    ca.slot_shift = slot_shift; ca.histo = d_histo; ca.histo_len = histo_len; ca.nbins = nbins;
    bool any = false;
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || !done[i]) continue;
        c->stats.fused_tasks++;
        outs[i].n = total[i];
        if (outs[i].n) {
            outs[i].entries = (u64 *)c->pool.alloc(outs[i].n * 16); outs[i].payoff = (u64 *)c->pool.alloc(outs[i].n * 8);
            if (!outs[i].entries || !outs[i].payoff) { rc = fail(c, HSK_ERR_OOM, "task output"); break; }
            ca.scratch_e[i] = a.t[i].scratch_e; ca.scratch_p[i] = a.t[i].scratch_p; ca.bounds[i] = a.t[i].bounds; ca.bin_off[i] = a.t[i].bin_cnt;
            ca.entries[i] = outs[i].entries; ca.payoff[i] = outs[i].payoff;
            any = true;
        }
    }
    if (any && rc == HSK_OK) hipLaunchKernelGGL(agg_ext_compact_kernel, dim3(256, AG_BATCH), dim3(AG_THREADS), 0, c->stream, ca);
    for (int i = 0; i < AG_BATCH && rc == HSK_OK; ++i) {
        if (bt[i].n == 0 || done[i]) continue;
        // the long way for this task: full-width passes (payload carried) from the current order, then the two-pass counter
        c->stats.redone_tasks++;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (own_scratch[i]) { c->pool.release(a.t[i].scratch_e); c->pool.release(a.t[i].scratch_p); own_scratch[i] = false; }
        const u64 payadd = pay_before[i];
        free_task_out(c, outs[i]);
        SortScratch sc1; rc = alloc_sort_scratch(c, sc1); if (rc) break;
        u64 *cur = bt[i].out_k, *other = (cur == bt[i].kA) ? bt[i].kB : bt[i].kA, *sk, *sv;
        u64 *vcur = bt[i].out_v, *vother = (vcur == bt[i].vA) ? bt[i].vB : bt[i].vA;
        rc = sort_task_device<1>(c, cur, other, vcur, vother, bt[i].n, K, sc1, &sk, &sv, false);
        free_sort_scratch(c, sc1);
        if (rc == HSK_OK) rc = count_task_device<1>(c, sk, sv, bt[i].n, payadd, d_histo, histo_len, outs[i]);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < AG_BATCH; ++i) if (own_scratch[i]) { c->pool.release(a.t[i].scratch_e); c->pool.release(a.t[i].scratch_p); }
    c->pool.release(d_bounds); c->pool.release(d_cnt); c->pool.release(d_flags);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Exchange / sort overlap (multi-GPU).  The owned tasks of every rank are cut into groups of
// XCD_BATCH consecutive tasks; group g+1 travels on `comm_stream` (RCCL send/recv, or device copies
// between the virtual ranks of the loopback driver) while group g is expanded, sorted and counted on
// the main stream.  The reference overlaps the same way with BATCH-sized MPI_Ialltoallv rounds
// (src/kmerops.cpp:130-196, exchange_supermer's stage loop); here the unit is a task group so that a
// sort batch never waits for bytes it does not need.  HSK_OVERLAP=0 selects one exchange up front.
// ------------------------------------------------------------------------------------------------
struct TaskInput { const u8 *len; BaseSource src; const u32 *pos; const int32_t *rid; };

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
    const SupermerStore *st = nullptr;
    const std::vector<SupermerStore> *st_all = nullptr;
    const std::vector<std::vector<ExchangePlan>> *pl_all = nullptr;     // [rank][group]
    u64 bytes_moved = 0;

    int plan(hsk_ctx *c_, int nranks_, int rank_, u32 ntasks, const std::vector<int32_t> &owner, const std::vector<u32> &order,
             const std::vector<u64> &M, const std::vector<u64> &task_base, std::vector<TaskSegs> &segs)
    {
        c = c_; nranks = nranks_; rank = rank_; ext = c->cfg.extension != 0;
        assign_task_groups(nranks, ntasks, owner, XCD_BATCH, group_of, ngroups);
        pl.resize(ngroups); xb.resize(ngroups); arrived.assign(ngroups, nullptr);
        segs.assign(ntasks, TaskSegs());
        for (int g = 0; g < ngroups; ++g) plan_exchange(nranks, rank, ntasks, owner, order, M, task_base, pl[g], segs, &group_of, g);
        return HSK_OK;
    }
    int post(int g)
    {
        ExchangeBuffers &b = xb[g]; const ExchangePlan &p = pl[g];
        b.len = (u8 *)c->pool.alloc(p.recv_tot_sup + 64); b.bytes = (u8 *)c->pool.alloc(p.recv_tot_bytes + 64); b.nbytes = p.recv_tot_bytes;
        if (ext) { b.pos = (u32 *)c->pool.alloc(p.recv_tot_sup * 4 + 64); b.rid = (int32_t *)c->pool.alloc(p.recv_tot_sup * 4 + 64); }
        if (!b.len || !b.bytes || (ext && (!b.pos || !b.rid))) return fail(c, HSK_ERR_OOM, "exchange buffers of group %d", g);
        // the pool hands out blocks whose previous user may still be running on the main stream: order the
        // transfer after everything launched there so far (that is the work of group g-2 and earlier)
        hipEvent_t fence = ev_get(c);
        HIPCHK(c, hipEventRecord(fence, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->comm_stream, fence, 0));
        ev_put(c, fence);
        hipStream_t s = c->comm_stream;
        if (st_all) {
            for (int src = 0; src < nranks; ++src) {
                const ExchangePlan &sp = (*pl_all)[src][g]; const SupermerStore &ss = (*st_all)[src];
                const u64 n = sp.send_sup[rank], nb = sp.send_bytes[rank];
                if (n != p.recv_sup[src] || nb != p.recv_bytes[src]) return fail(c, HSK_ERR_INTERNAL, "exchange plan mismatch %d->%d (group %d)", src, rank, g);
                if (!n) continue;
                HIPCHK(c, hipMemcpyAsync(b.len + p.recv_sup_off[src], ss.sm_len + sp.send_sup_off[rank], n, hipMemcpyDeviceToDevice, s));
                HIPCHK(c, hipMemcpyAsync(b.bytes + p.recv_byte_off[src], ss.sm_bytes + sp.send_byte_off[rank], nb, hipMemcpyDeviceToDevice, s));
                if (ext) {
                    HIPCHK(c, hipMemcpyAsync(b.pos + p.recv_sup_off[src], ss.sm_pos + sp.send_sup_off[rank], n * 4, hipMemcpyDeviceToDevice, s));
                    HIPCHK(c, hipMemcpyAsync(b.rid + p.recv_sup_off[src], ss.sm_rid + sp.send_sup_off[rank], n * 4, hipMemcpyDeviceToDevice, s));
                }
            }
        } else {
            int rc = post_exchange(c->comm, s, ext, p, st->sm_len, st->sm_bytes, st->sm_pos, st->sm_rid, b);
            if (rc) return fail(c, HSK_ERR_COMM, "supermer exchange (group %d) failed: %d (%s)", g, rc, c->comm.last_error.c_str());
        }
        bytes_moved += p.recv_tot_bytes + p.recv_tot_sup * (ext ? 9 : 1);
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
            if (arrived[released]) { ev_put(c, arrived[released]); arrived[released] = nullptr; }
        }
    }
    // every rank must take part in every group even when it owns no task of it
    int finish()
    {
        while (posted < ngroups) { int rc = post(posted); if (rc) return rc; ++posted; }
        HIPCHK(c, hipStreamSynchronize(c->comm_stream));
        release_below(ngroups);
        return HSK_OK;
    }
    TaskInput input(u32 t) const
    {
        const ExchangeBuffers &b = xb[group_of[t]];
        TaskInput in; in.len = b.len; in.src = source_from_bytes(b.bytes, b.nbytes); in.pos = b.pos; in.rid = b.rid;
        return in;
    }
};

static bool overlap_enabled()
{
    static const bool on = !(getenv("HSK_OVERLAP") && atoi(getenv("HSK_OVERLAP")) == 0);
    return on;
}

// ---- heavy-hitter tasks (a8): the owner's side --------------------------------------------------------------
// d_entries: the {k-mer, count} lists of all ranks for one task, concatenated (n entries, each list key-ordered,
// a key at most once per list).  Orders them by key with the count as payload, sums equal keys, filters [L, U].
static int heavy_merge_task(hsk_ctx *c, const u64 *d_entries, u64 n, u64 *d_histo, u32 histo_len, TaskOut &out)
{
    out = TaskOut();
    if (n == 0) return HSK_OK;
    u64 *kA, *kB, *vA, *vB;
    DALLOC(c, kA, u64 *, n * 8 + 64); DALLOC(c, kB, u64 *, n * 8 + 64); DALLOC(c, vA, u64 *, n * 8 + 64); DALLOC(c, vB, u64 *, n * 8 + 64);
    hipLaunchKernelGGL(heavy_split_kernel, dim3((u32)std::min<u64>((n + HV_THREADS - 1) / HV_THREADS, 4096)), dim3(HV_THREADS), 0, c->stream, d_entries, n, kA, vA);
    SortScratch sc; int rc = alloc_sort_scratch(c, sc); if (rc) return rc;
    u64 *sk, *sv;
    rc = sort_task_device<1>(c, kA, kB, vA, vB, n, c->cfg.kmer_size, sc, &sk, &sv);
    free_sort_scratch(c, sc);
    if (rc) return rc;
    const u64 ntiles = (n + HV_THREADS - 1) / HV_THREADS;
    u64 *d_tile, *d_total;
    DALLOC(c, d_tile, u64 *, ntiles * 8 + 64); DALLOC(c, d_total, u64 *, 256);
    HeavyMergeArgs a; memset(&a, 0, sizeof a);
    a.keys = sk; a.cnts = sv; a.n = n; a.lower = (u64)c->cfg.lower_freq; a.upper = (u64)c->cfg.upper_freq; a.tile_cnt = d_tile; a.histo = d_histo; a.histo_len = histo_len;
    hipLaunchKernelGGL(heavy_merge_kernel<false>, dim3((u32)ntiles), dim3(HV_THREADS), 0, c->stream, a);
    hipLaunchKernelGGL(count_scan_kernel, dim3(1), dim3(CNT_THREADS), 0, c->stream, d_tile, ntiles, d_total);
    u64 *tot = (u64 *)((char *)c->pinned + c->pinned_bytes - 128);
    HIPCHK(c, hipMemcpyAsync(tot, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out.n = tot[0];
    if (out.n) {
        DALLOC(c, out.entries, u64 *, out.n * 16);
        a.entries = out.entries;
        hipLaunchKernelGGL(heavy_merge_kernel<true>, dim3((u32)ntiles), dim3(HV_THREADS), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pool.release(kA); c->pool.release(kB); c->pool.release(vA); c->pool.release(vB); c->pool.release(d_tile); c->pool.release(d_total);
    return HSK_OK;
}

struct HeavyIn { u32 task; u64 *d_entries; u64 n; };       // a heavy task this rank owns: concatenated lists of all ranks
struct ProcExtra {
    bool force_batch = false;                              // every task through the batch kernels (partial batches padded)
    const std::vector<HeavyIn> *heavy_in = nullptr;        // merged and filtered here (they have no supermers)
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

    // ---- per task: expand, sort, count ---------------------------------------------------------------
    const u32 histo_len = (u32)std::min<int64_t>((int64_t)c->cfg.upper_freq + 1, 65536);    // (U <= 65535 except in the unfiltered pre-aggregation)
    u64 *d_histo; DALLOC(c, d_histo, u64 *, (size_t)histo_len * 8);
    HIPCHK(c, hipMemsetAsync(d_histo, 0, (size_t)histo_len * 8, c->stream));
    // Tasks are sorted eight at a time, one per XCD (sort_batch_device); a remainder of fewer than eight
    // tasks goes through the single-task kernel.  HSK_XCD_BATCH=0 forces the single-task path.
    static const bool batch_enabled = !(getenv("HSK_XCD_BATCH") && atoi(getenv("HSK_XCD_BATCH")) == 0);
    std::vector<u32> mine;
    for (u32 t = 0; t < ntasks; ++t) if (owner[t] == rank && segs[t].nkmers) mine.push_back(t);
    // A remainder of three or more tasks is padded to a full batch with empty slots (an XCD without a task idles, which
    // still beats eight full-width passes per task on the single-task path); ex->force_batch pads any remainder.
    const u32 EMPTY_TASK = ~0u;
    TaskSegs empty_segs;
    std::vector<TaskOut> touts(ntasks);
    std::vector<u32> mine_done;                          // tasks finished by the one-pass loop below
    // ---- the one-pass plan (HSK_ONEPASS=1): batches of up to 64 small tasks -------------------------------------
    // expand (8 tasks per launch, digit histogram of the top 8 bits) -> ONE scatter pass over all tasks of the batch in
    // one launch -> aggregation over 8-bit prefix bins (8 tasks per launch); tasks whose bins overflow the LDS table are
    // ordered on the next 8 bits too and finished over 16-bit bins.  Single GPU only (the exchange feeds groups of 8).
    if constexpr (NW == 1) {
        const bool op = !ext && !feeder && batch_enabled && onepass_enabled() && hybrid_enabled() && finish_enabled() && agg_enabled() &&
                        max_task <= ONEPASS_MAX_TASK && !mine.empty() && !(ex && ex->heavy_in && !ex->heavy_in->empty());
        if (op) {
            const int nbmax = (int)std::min<size_t>(MANY_MAX, (mine.size() + 7) / 8 * 8);
            std::vector<u64 *> kAm(nbmax, nullptr), kBm(nbmax, nullptr);
            for (int i = 0; i < nbmax; ++i) { DALLOC(c, kAm[i], u64 *, max_task * 8 + 64); DALLOC(c, kBm[i], u64 *, max_task * 8 + 64); }
            u64 *d_gh; DALLOC(c, d_gh, u64 *, (size_t)nbmax * MAX_PASSES * 256 * 8);
            PassDesc plan1[MAX_PASSES];
            const int np1 = make_hybrid_plan(plan1, 8, 0);
            TaskInput dflt1; dflt1.len = x_len; dflt1.src = x_src; dflt1.pos = x_pos; dflt1.rid = x_rid;
            for (size_t mb = 0; mb < mine.size(); mb += MANY_MAX) {
                const int nreal = (int)std::min<size_t>(MANY_MAX, mine.size() - mb);
                const int nb = (nreal + 7) / 8 * 8;
                BatchTask bt[MANY_MAX];
                pt.begin(PH_EXTRACT);
                HIPCHK(c, hipMemsetAsync(d_gh, 0, (size_t)nb * MAX_PASSES * 256 * 8, c->stream));
                for (int c0 = 0; c0 < nb; c0 += XCD_BATCH) {
                    ExpandJob jobs[XCD_BATCH];
                    for (int i = 0; i < XCD_BATCH; ++i) {
                        BatchTask &b = bt[c0 + i]; b = BatchTask(); b.kA = kAm[c0 + i]; b.kB = kBm[c0 + i];
                        jobs[i] = ExpandJob(); jobs[i].ts = &empty_segs;
                        if (c0 + i >= nreal) continue;
                        const u32 t = mine[mb + c0 + i];
                        b.n = segs[t].nkmers;
                        jobs[i].ts = &segs[t]; jobs[i].sm_len = dflt1.len; jobs[i].src = dflt1.src; jobs[i].sm_pos = dflt1.pos; jobs[i].sm_rid = dflt1.rid;
                        jobs[i].keys = b.kA; jobs[i].vals = nullptr; jobs[i].ghist = d_gh + (size_t)(c0 + i) * MAX_PASSES * 256;
                    }
                    int rc = expand_batch<NW>(c, jobs, XCD_BATCH, np1, plan1); if (rc) return rc;
                }
                pt.end(PH_EXTRACT);
                pt.begin(PH_SORT);
                { int rc = sort_many_onepass<NW>(c, bt, nb, d_gh); if (rc) return rc; }
                pt.end(PH_SORT);
                pt.begin(PH_COUNT);
                for (int c0 = 0; c0 < nb; c0 += XCD_BATCH) {
                    TaskOut fo[XCD_BATCH];
                    int rc = agg_finish_batch_device<1>(c, bt + c0, K, max_task, d_histo, histo_len, fo, 8); if (rc) return rc;
                    BatchTask b2[XCD_BATCH]; bool any_miss = false;
                    for (int i = 0; i < XCD_BATCH; ++i) {
                        b2[i] = BatchTask();
                        if (!fo[i].failed) continue;
                        any_miss = true; c->stats.onepass_misses++;
                        b2[i].n = bt[c0 + i].n; b2[i].kA = bt[c0 + i].out_k; b2[i].kB = (bt[c0 + i].out_k == bt[c0 + i].kA) ? bt[c0 + i].kB : bt[c0 + i].kA;
                    }
                    if (any_miss) {
                        rc = sort_batch_device<NW>(c, b2, K, true, AG_PREFIX_BITS, nullptr); if (rc) return rc;
                        TaskOut f2[XCD_BATCH];
                        rc = agg_finish_batch_device<1>(c, b2, K, max_task, d_histo, histo_len, f2, AG_PREFIX_BITS); if (rc) return rc;
                        for (int i = 0; i < XCD_BATCH; ++i) if (fo[i].failed) fo[i] = f2[i];
                    }
                    for (int i = 0; i < XCD_BATCH; ++i) if (c0 + i < nreal) touts[mine[mb + c0 + i]] = fo[i];
                }
                pt.end(PH_COUNT);
            }
            for (int i = 0; i < nbmax; ++i) { c->pool.release(kAm[i]); c->pool.release(kBm[i]); }
            c->pool.release(d_gh);
            mine_done.swap(mine);                            // nothing left for the two-pass loops
        }
    }
    const bool forced = ex && ex->force_batch && batch_enabled;
    if ((batch_enabled && mine.size() >= (size_t)XCD_BATCH && mine.size() % XCD_BATCH >= 3) || (forced && !mine.empty()))
        while (mine.size() % XCD_BATCH) mine.push_back(EMPTY_TASK);
    const bool batch = batch_enabled && mine.size() >= (size_t)XCD_BATCH;
    const int nsets = batch ? XCD_BATCH : 1;
    // Two batches in flight (single GPU): batch b+1 is expanded on the second stream while batch b is sorted and
    // counted on the main stream.  The expand kernel waits on memory latency for most of its life, the radix passes
    // are bandwidth-bound and the aggregation is issue-bound: side by side they fill each other's gaps.  Every
    // buffer the second stream touches is allocated up front (the pool's reuse rule is per stream).
    // (off unless HSK_PIPELINE=1: the gain is ~1.5 % and every per-kernel duration, hence the reported roofline of the
    // scatter pass, is inflated by whatever runs beside it)
    static const bool pipe_enabled = getenv("HSK_PIPELINE") && atoi(getenv("HSK_PIPELINE")) != 0;
    const bool piped = batch && !feeder && pipe_enabled && mine.size() >= 2 * (size_t)XCD_BATCH;
    const int nslot = piped ? 2 : 1;
    u64 *kAs[2][XCD_BATCH] = {{nullptr}}, *kBs[2][XCD_BATCH] = {{nullptr}}, *vAs[2][XCD_BATCH] = {{nullptr}}, *vBs[2][XCD_BATCH] = {{nullptr}};
    u64 **kA = kAs[0], **kB = kBs[0], **vA = vAs[0], **vB = vBs[0];          // slot 0: also the single-task path
    SortScratch sc;
    if (max_task) {
        for (int sl = 0; sl < nslot; ++sl) for (int i = 0; i < nsets; ++i) {
            DALLOC(c, kAs[sl][i], u64 *, max_task * NW * 8 + 64); DALLOC(c, kBs[sl][i], u64 *, max_task * NW * 8 + 64);
            if (ext) { DALLOC(c, vAs[sl][i], u64 *, max_task * 8 + 64); DALLOC(c, vBs[sl][i], u64 *, max_task * 8 + 64); }
        }
        int rc = alloc_sort_scratch(c, sc); if (rc) return rc;
    }
    u64 *d_ghist_slot[2] = {nullptr, nullptr};
    ExpandScratch xpre[2][XCD_BATCH];
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    bool done_valid[2] = {false, false};
    hipStream_t xstream = piped ? c->comm_stream : c->stream;
    if (batch) for (int sl = 0; sl < nslot; ++sl) DALLOC(c, d_ghist_slot[sl], u64 *, (size_t)XCD_BATCH * MAX_PASSES * 256 * 8);
    if (piped) {
        u64 max_tiles = 0; size_t max_seg = 1;
        for (u32 t : mine) { if (t == EMPTY_TASK) continue; max_tiles = std::max(max_tiles, segs[t].ntiles); max_seg = std::max(max_seg, segs[t].segs.size()); }
        for (int sl = 0; sl < 2; ++sl) {
            for (int i = 0; i < XCD_BATCH; ++i) {
                DALLOC(c, xpre[sl][i].d_segs, ExpSeg *, sizeof(ExpSeg) * max_seg);
                DALLOC(c, xpre[sl][i].d_tile_sum, u64 *, max_tiles * 16 + 64);
                DALLOC(c, xpre[sl][i].d_tile_off, u64 *, max_tiles * 16 + 64);
            }
            ev_ready[sl] = ev_get(c); ev_done[sl] = ev_get(c);
        }
        // everything the pool handed out above may still be in use by earlier main-stream work
        hipEvent_t fence = ev_get(c);
        HIPCHK(c, hipEventRecord(fence, c->stream));
        HIPCHK(c, hipStreamWaitEvent(xstream, fence, 0));
        ev_put(c, fence);
    }
    u64 n_total = 0, pay_total = 0;
    // payload offsets are global over the owned tasks in ascending id: prefix of k-mer counts
    std::vector<u64> pay_before(ntasks, 0);
    { u64 acc = 0; for (u32 t : mine) { if (t == EMPTY_TASK) continue; pay_before[t] = acc; if (ext) acc += segs[t].nkmers; } }
    TaskInput dflt; dflt.len = x_len; dflt.src = x_src; dflt.pos = x_pos; dflt.rid = x_rid;
    // fused finish: one-word keys (aggregating or tile finish), two-word keys with K >= 40 (aggregating finish only)
    const bool fused = !ext && finish_enabled() && (NW == 1 ? hybrid_enabled() : (NW == 2 && agg_enabled() && prefix_plan_ok<NW>(K, true)));
    const bool agg = fused && agg_enabled();
    // EXTENSION with one-word keys: two passes on the top 16 bits (payload carried) + grouping aggregation
    const bool fused_ext = ext && NW == 1 && hybrid_enabled() && finish_enabled() && agg_enabled();
    int slot_prefix[2] = {0, 0};                          // the digit plan a slot's batch was expanded for
    BatchTask bts[2][XCD_BATCH];
    // one launch expands the eight tasks mine[bpos ..] into the slot's buffers and counts the digits of the passes that follow
    auto issue_expand = [&](size_t bpos, int sl) -> int {
        const int prefix_bits = (agg || fused_ext) ? AG_PREFIX_BITS : 64 - HYBRID_SHIFT;
        slot_prefix[sl] = prefix_bits;
        PassDesc plan[MAX_PASSES];
        const int npass = batch_pass_plan<NW>(c, K, fused || fused_ext, prefix_bits, plan);
        if (piped && done_valid[sl]) HIPCHK(c, hipStreamWaitEvent(xstream, ev_done[sl], 0));     // the slot's previous batch is counted
        pt.begin(PH_EXTRACT, xstream);
        HIPCHK(c, hipMemsetAsync(d_ghist_slot[sl], 0, (size_t)XCD_BATCH * MAX_PASSES * 256 * 8, xstream));
        ExpandJob jobs[XCD_BATCH];
        for (int i = 0; i < XCD_BATCH; ++i) {
            const u32 t = mine[bpos + i];
            BatchTask &b = bts[sl][i];
            b = BatchTask();
            b.kA = kAs[sl][i]; b.kB = kBs[sl][i]; b.vA = vAs[sl][i]; b.vB = vBs[sl][i];
            if (t == EMPTY_TASK) { jobs[i] = ExpandJob(); jobs[i].ts = &empty_segs; continue; }
            b.n = segs[t].nkmers;
            const TaskInput in = feeder ? feeder->input(t) : dflt;
            jobs[i].ts = &segs[t]; jobs[i].sm_len = in.len; jobs[i].src = in.src; jobs[i].sm_pos = in.pos; jobs[i].sm_rid = in.rid;
            jobs[i].keys = b.kA; jobs[i].vals = b.vA; jobs[i].ghist = d_ghist_slot[sl] + (size_t)i * MAX_PASSES * 256;
        }
        int rc = expand_batch<NW>(c, jobs, XCD_BATCH, npass, plan, xstream, piped ? xpre[sl] : nullptr); if (rc) return rc;
        pt.end(PH_EXTRACT, xstream);
        if (piped) HIPCHK(c, hipEventRecord(ev_ready[sl], xstream));
        return HSK_OK;
    };
    size_t pos = 0;
    const size_t nbatch = batch ? mine.size() / XCD_BATCH : 0;
    if (piped) { int rc = issue_expand(0, 0); if (rc) return rc; }
    for (size_t b = 0; b < nbatch; ++b, pos += XCD_BATCH) {
        const int sl = piped ? (int)(b & 1) : 0;
        if (feeder) {                                   // exposed (not overlapped) part of the exchange
            pt.begin(PH_EXCH);
            for (int i = 0; i < XCD_BATCH; ++i) { if (mine[pos + i] == EMPTY_TASK) continue; int rc = feeder->need(feeder->group_of[mine[pos + i]]); if (rc) return rc; }
            pt.end(PH_EXCH);
        }
        if (!piped) { int rc = issue_expand(pos, 0); if (rc) return rc; }
        else {
            if (b + 1 < nbatch) { int rc = issue_expand(pos + XCD_BATCH, (int)((b + 1) & 1)); if (rc) return rc; }
            HIPCHK(c, hipStreamWaitEvent(c->stream, ev_ready[sl], 0));
        }
        BatchTask *bt = bts[sl];
        if (feeder) feeder->release_below((pos + XCD_BATCH < mine.size() && mine[pos + XCD_BATCH] != EMPTY_TASK) ? feeder->group_of[mine[pos + XCD_BATCH]] : feeder->ngroups);
        const int prefix_bits = slot_prefix[sl];
        pt.begin(PH_SORT);
        { int rc = sort_batch_device<NW>(c, bt, K, fused || fused_ext, prefix_bits, d_ghist_slot[sl]); if (rc) return rc; }
        pt.end(PH_SORT);
        pt.begin(PH_COUNT);
        if (fused_ext) {
            if constexpr (NW == 1) {
                TaskOut fo[XCD_BATCH]; u64 pb[XCD_BATCH];
                for (int i = 0; i < XCD_BATCH; ++i) pb[i] = mine[pos + i] != EMPTY_TASK ? pay_before[mine[pos + i]] : 0;
                int rc = agg_ext_finish_batch_device(c, bt, K, pb, d_histo, histo_len, fo); if (rc) return rc;
                for (int i = 0; i < XCD_BATCH; ++i) if (mine[pos + i] != EMPTY_TASK) touts[mine[pos + i]] = fo[i];
            }
        } else if (fused) {
            if constexpr (NW <= 2) {
                TaskOut fo[XCD_BATCH];
                int rc;
                if constexpr (NW == 1) rc = agg ? agg_finish_batch_device<1>(c, bt, K, max_task, d_histo, histo_len, fo, prefix_bits)
                                                : finish_batch_device<1>(c, bt, K, max_task, d_histo, histo_len, fo);
                else rc = agg_finish_batch_device<NW>(c, bt, K, max_task, d_histo, histo_len, fo, prefix_bits);
                if (rc) return rc;
                for (int i = 0; i < XCD_BATCH; ++i) if (mine[pos + i] != EMPTY_TASK) touts[mine[pos + i]] = fo[i];
            }
        } else {
            for (int i = 0; i < XCD_BATCH; ++i) {
                const u32 t = mine[pos + i];
                if (t == EMPTY_TASK) continue;
                int rc = count_task_device<NW>(c, bt[i].out_k, bt[i].out_v, bt[i].n, pay_before[t], d_histo, histo_len, touts[t]); if (rc) return rc;
            }
        }
        pt.end(PH_COUNT);
        if (piped) { HIPCHK(c, hipEventRecord(ev_done[sl], c->stream)); done_valid[sl] = true; }
    }
    if (piped) {
        HIPCHK(c, hipStreamSynchronize(xstream));
        for (int sl = 0; sl < 2; ++sl) {
            for (int i = 0; i < XCD_BATCH; ++i) expand_release(c, xpre[sl][i]);
            ev_put(c, ev_ready[sl]); ev_put(c, ev_done[sl]);
        }
    }
    for (; pos < mine.size(); ++pos) {
        const u32 t = mine[pos];
        const u64 n = segs[t].nkmers;
        int rc;
        if (feeder) { pt.begin(PH_EXCH); rc = feeder->need(feeder->group_of[t]); pt.end(PH_EXCH); if (rc) return rc; }
        pt.begin(PH_EXTRACT);
        const TaskInput in = feeder ? feeder->input(t) : dflt;
        rc = expand_task<NW>(c, segs[t], in.len, in.src, in.pos, in.rid, kA[0], vA[0]); if (rc) return rc;
        pt.end(PH_EXTRACT);
        if (feeder) feeder->release_below(pos + 1 < mine.size() ? feeder->group_of[mine[pos + 1]] : feeder->ngroups);
        pt.begin(PH_SORT);
        u64 *sk, *sv;
        rc = sort_task_device<NW>(c, kA[0], kB[0], vA[0], vB[0], n, K, sc, &sk, &sv); if (rc) return rc;
        pt.end(PH_SORT);
        pt.begin(PH_COUNT);
        rc = count_task_device<NW>(c, sk, sv, n, pay_before[t], d_histo, histo_len, touts[t]); if (rc) return rc;
        pt.end(PH_COUNT);
    }
    // heavy-hitter tasks this rank owns arrive as k-mer lists: order, sum, filter
    if (ex && ex->heavy_in) {
        pt.begin(PH_COUNT);
        for (const HeavyIn &hv : *ex->heavy_in) {
            if constexpr (NW == 1) { int rc = heavy_merge_task(c, hv.d_entries, hv.n, d_histo, histo_len, touts[hv.task]); if (rc) return rc; }
            mine.push_back(hv.task);
        }
        pt.end(PH_COUNT);
    }
    for (u32 t : mine) { if (t == EMPTY_TASK) continue; touts[t].pay_base = pay_before[t]; n_total += touts[t].n; pay_total += touts[t].npay; }
    for (u32 t : mine_done) { n_total += touts[t].n; pay_total += touts[t].npay; }
    if (feeder) { int rc = feeder->finish(); if (rc) return rc; }
    {
        int rc = check_device_error(c); if (rc) return rc;
    }
    for (int sl = 0; sl < nslot; ++sl) for (int i = 0; i < nsets; ++i) { c->pool.release(kAs[sl][i]); c->pool.release(kBs[sl][i]); c->pool.release(vAs[sl][i]); c->pool.release(vBs[sl][i]); }
    free_sort_scratch(c, sc);
    c->pool.release(d_ghist_slot[0]); c->pool.release(d_ghist_slot[1]);

    // ---- result ----------------------------------------------------------------------------------------
    pt.begin(PH_D2H);
    out->n = n_total;
    out->task_off = (uint64_t *)host_alloc(rp, (size_t)(ntasks + 1) * 8);
    out->histo = (uint64_t *)host_alloc(rp, (size_t)histo_len * 8);
    out->histo_len = histo_len;
    if (!out->task_off || !out->histo) return fail(c, HSK_ERR_OOM, "pinned host allocation failed");
    HIPCHK(c, hipMemcpyAsync(out->histo, d_histo, (size_t)histo_len * 8, hipMemcpyDeviceToHost, c->stream));
    const bool keep = (c->cfg.flags & HSK_FLAG_KEEP_DEVICE) != 0;
    if (!keep) {
        out->entries = (uint64_t *)host_alloc(rp, n_total * (NW + 1) * 8);
        if (!out->entries) return fail(c, HSK_ERR_OOM, "pinned host allocation of %llu bytes failed", (unsigned long long)(n_total * (NW + 1) * 8));
        if (ext) {
            out->payload_off = (uint64_t *)host_alloc(rp, (n_total + 1) * 8);
            out->pos = (uint32_t *)host_alloc(rp, pay_total * 4);
            out->rid = (int32_t *)host_alloc(rp, pay_total * 4);
            if (!out->payload_off || !out->pos || !out->rid) return fail(c, HSK_ERR_OOM, "pinned host allocation failed");
        }
    }
    u64 o = 0, po = 0;
    for (u32 t = 0; t < ntasks; ++t) {
        out->task_off[t] = o;
        TaskOut &to = touts[t];
        if (!keep) {
            if (to.n) HIPCHK(c, hipMemcpyAsync(out->entries + o * (NW + 1), to.entries, to.n * (NW + 1) * 8, hipMemcpyDeviceToHost, c->stream));
            if (ext && to.n) HIPCHK(c, hipMemcpyAsync(out->payload_off + o, to.payoff, to.n * 8, hipMemcpyDeviceToHost, c->stream));
            if (ext && to.npay) {
                HIPCHK(c, hipMemcpyAsync(out->pos + po, to.pos, to.npay * 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(out->rid + po, to.rid, to.npay * 4, hipMemcpyDeviceToHost, c->stream));
            }
        }
        o += to.n; po += to.npay;
    }
    out->task_off[ntasks] = o;
    pt.end(PH_D2H);
    if (pt_total_open) pt.end(PH_TOTAL);
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
// HeavyHitterClassifier (reference src/kmerops.cpp:1157-1199) on the GLOBAL k-mer counts; forced plain with
// EXTENSION or PLAIN_CLASSIFIER (kmerops.cpp:109-113) and, here, for keys of more than one word.
static bool heavy_enabled(hsk_ctx *c, int nw, int nranks)
{
    static const bool env_on = !(getenv("HSK_HEAVY") && atoi(getenv("HSK_HEAVY")) == 0);
    return env_on && nranks > 1 && nw == 1 && c->cfg.extension == 0 && (c->cfg.flags & HSK_FLAG_PLAIN_CLASSIFIER) == 0 &&
           hybrid_enabled() && finish_enabled() && agg_enabled();
}
static double heavy_ratio() { static const double r = getenv("HSK_UNBALANCED_RATIO") ? atof(getenv("HSK_UNBALANCED_RATIO")) : 2.3; return r; }

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
    rc = process_rank<NW>(c, ntasks, own, 0, segs, sth.sm_len, source_from_packed(d_packed, packed_bytes, sth.sm_gpos), nullptr, nullptr, &tmp, rp, pt, false, nullptr, &ex);
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
    PhaseTimer pt(c);
    pt.begin(PH_TOTAL);

    u32 ntasks = c->cfg.ntasks ? (u32)c->cfg.ntasks : auto_ntasks(c, packed_bytes, nranks);
    if (c->comm.active() && !c->cfg.ntasks) {
        // every rank must use the same task count: take the maximum of the local proposals
        u64 v = ntasks; int rc = c->comm.allreduce_max_u64(&v, 1, c->stream, c->pool); if (rc) return fail(c, HSK_ERR_COMM, "allreduce(ntasks) failed: %d", rc);
        ntasks = (u32)v;
    }
    out->ntasks = (int32_t)ntasks;
    std::vector<int32_t> owner(ntasks, 0);
    std::vector<u32> order(ntasks);
    for (u32 t = 0; t < ntasks; ++t) order[t] = t;

    // ---- parse ------------------------------------------------------------------------------------
    SupermerStore st;
    std::vector<u8> is_heavy(ntasks, 0);
    std::vector<TaskOut> hlists;                         // this rank's {k-mer, count} lists of the heavy tasks
    std::vector<HeavyIn> hin;                            // heavy tasks this rank owns: the lists of all ranks
    bool any_heavy = false;
    pt.begin(PH_PARSE);
    {
        // the reads are hashed once (parse_count); multi-GPU: the dispatcher needs the global task sizes
        // before the storage order (tasks grouped by owner rank) is known, then parse_place lays the supermers out
        ParseJob job;
        int rc = parse_count(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, ntasks, job);
        if (rc) { parse_release(c, job); return rc; }
        if (nranks > 1) {
            std::vector<u64> bytes(ntasks);
            for (u32 t = 0; t < ntasks; ++t) bytes[t] = job.task_tot[3 * t + 1] + job.task_tot[3 * t] * (ext ? 9 : 1);
            // heavy-hitter tasks (a8): classified on the global k-mer counts; every rank pre-aggregates its own share
            if (heavy_enabled(c, NW, nranks)) {
                std::vector<u64> kg(ntasks); std::vector<int32_t> types(ntasks, 0);
                for (u32 t = 0; t < ntasks; ++t) kg[t] = job.task_tot[3 * t + 2];
                rc = c->comm.allreduce_sum_u64(kg.data(), ntasks, c->stream, c->pool);
                if (rc) { parse_release(c, job); return fail(c, HSK_ERR_COMM, "allreduce(task k-mers) failed: %d", rc); }
                plan_classify(kg.data(), (int)ntasks, heavy_ratio(), types.data());
                for (u32 t = 0; t < ntasks; ++t) if (types[t] == 1) { is_heavy[t] = 1; any_heavy = true; }
            }
            if (any_heavy) {
                std::vector<u8> failed;
                rc = heavy_preaggregate<NW>(c, job, d_packed, packed_bytes, is_heavy, hlists, failed);
                if (rc) { parse_release(c, job); return rc; }
                std::vector<u64> bad(ntasks);
                for (u32 t = 0; t < ntasks; ++t) bad[t] = failed[t];
                rc = c->comm.allreduce_max_u64(bad.data(), ntasks, c->stream, c->pool);     // a task one rank could not aggregate travels as supermers everywhere
                if (rc) { parse_release(c, job); return fail(c, HSK_ERR_COMM, "allreduce(heavy flags) failed: %d", rc); }
                any_heavy = false;
                for (u32 t = 0; t < ntasks; ++t) {
                    if (!is_heavy[t]) continue;
                    if (bad[t]) { is_heavy[t] = 0; free_task_out(c, hlists[t]); continue; }
                    any_heavy = true; c->stats.heavy_tasks++;
                    bytes[t] = hlists[t].n * (u64)(NW + 1) * 8;                               // ScatteredKmerList::get_size_bytes
                }
            }
            rc = c->comm.allreduce_sum_u64(bytes.data(), ntasks, c->stream, c->pool);
            if (rc) { parse_release(c, job); return fail(c, HSK_ERR_COMM, "allreduce(task sizes) failed: %d", rc); }
            rc = plan_dispatch(bytes.data(), (int)ntasks, nranks, c->cfg.plain_dispatcher != 0, c->cfg.dispatch_upper_coe, c->cfg.dispatch_step, owner.data());
            if (rc) { parse_release(c, job); return fail(c, HSK_ERR_DISPATCH, "%s", hsk_strerror(HSK_ERR_DISPATCH)); }
            std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return owner[x] < owner[y]; });
        }
        rc = parse_place(c, job, order, st, any_heavy ? &is_heavy : nullptr);
        parse_release(c, job);
        if (rc) return rc;
    }
    pt.end(PH_PARSE);
    out->total_supermers = st.tot_sup; out->total_supermer_bytes = st.tot_bytes + st.tot_sup * (ext ? 9 : 1);

    // ---- exchange (multi-GPU) ---------------------------------------------------------------------
    // After this block `segs[t]` lists where the supermers of owned task t live.
    std::vector<TaskSegs> segs(ntasks);
    const u8 *x_len = st.sm_len; const u32 *x_pos = st.sm_pos; const int32_t *x_rid = st.sm_rid;
    BaseSource x_src = source_from_packed(d_packed, packed_bytes, st.sm_gpos);
    ExchangeBuffers xb;
    GroupFeeder feeder; bool fed = false;
    pt.begin(PH_EXCH);
    if (nranks > 1) {
        int rc = pack_store_bytes(c, st, x_src); if (rc) return rc;
        if (overlap_enabled()) {
            // size matrix: every rank contributes its row, the sum is the full matrix
            std::vector<u64> M((size_t)nranks * ntasks * 3, 0);
            for (size_t i = 0; i < (size_t)ntasks * 3; ++i) M[(size_t)rank * ntasks * 3 + i] = st.task_tot[i];
            rc = c->comm.allreduce_sum_u64(M.data(), M.size(), c->stream, c->pool);
            if (rc) return fail(c, HSK_ERR_COMM, "allreduce(size matrix) failed: %d (%s)", rc, c->comm.last_error.c_str());
            rc = feeder.plan(c, nranks, rank, ntasks, owner, order, M, st.task_base, segs); if (rc) return rc;
            feeder.st = &st; fed = true;
        } else {
            rc = exchange_supermers(c->comm, c->stream, c->pool, ext, K, ntasks, owner, order, st.task_tot, st.task_base,
                                    st.sm_len, st.sm_bytes, st.sm_pos, st.sm_rid, xb, segs);
            if (rc) return fail(c, HSK_ERR_COMM, "supermer exchange failed: %d (%s)", rc, c->comm.last_error.c_str());
            x_len = xb.len; x_pos = xb.pos; x_rid = xb.rid;
            x_src = source_from_bytes(xb.bytes, xb.nbytes);
            free_store(c, st);
        }
    } else {
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
        for (size_t i = 0; i < nh; ++i) {
            const u32 t = hv_tasks[i];
            if (owner[t] != rank) continue;
            HeavyIn hv; hv.task = t; hv.n = 0; hv.d_entries = nullptr;
            for (int p = 0; p < nranks; ++p) hv.n += Hn[(size_t)p * nh + i];
            if (hv.n) DALLOC(c, hv.d_entries, u64 *, hv.n * ew);
            hin.push_back(hv);
        }
        Comm &cm = c->comm;
        if ((rc = cm.check(cm.api->GroupStart(), "ncclGroupStart"))) return fail(c, HSK_ERR_COMM, "%s", cm.last_error.c_str());
        size_t hi = 0;
        for (size_t i = 0; i < nh && rc == 0; ++i) {
            const u32 t = hv_tasks[i];
            if (owner[t] == rank) {
                HeavyIn &hv = hin[hi++];
                u64 o = 0;
                for (int p = 0; p < nranks && rc == 0; ++p) {
                    const u64 n = Hn[(size_t)p * nh + i];
                    if (n && p != rank) rc = cm.check(cm.api->Recv((char *)hv.d_entries + o * ew, n * ew, RCCL_UINT8, p, cm.comm, c->stream), "ncclRecv(heavy list)");
                    o += n;
                }
            } else if (hlists[t].n) {
                rc = cm.check(cm.api->Send(hlists[t].entries, hlists[t].n * ew, RCCL_UINT8, owner[t], cm.comm, c->stream), "ncclSend(heavy list)");
            }
        }
        const int rc2 = cm.check(cm.api->GroupEnd(), "ncclGroupEnd");
        if (rc || rc2) return fail(c, HSK_ERR_COMM, "heavy-hitter list exchange failed: %s", cm.last_error.c_str());
        hi = 0;
        for (size_t i = 0; i < nh; ++i) {                                   // own share: device copy
            const u32 t = hv_tasks[i];
            if (owner[t] != rank) continue;
            HeavyIn &hv = hin[hi++];
            u64 o = 0; for (int p = 0; p < rank; ++p) o += Hn[(size_t)p * nh + i];
            if (hlists[t].n) HIPCHK(c, hipMemcpyAsync((char *)hv.d_entries + o * ew, hlists[t].entries, hlists[t].n * ew, hipMemcpyDeviceToDevice, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (auto &to : hlists) free_task_out(c, to);
    }
    pt.end(PH_EXCH);
    ProcExtra ex; ex.heavy_in = &hin;
    int rc = process_rank<NW>(c, ntasks, owner, rank, segs, x_len, x_src, x_pos, x_rid, out, rp, pt, true, fed ? &feeder : nullptr, &ex);
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
    // 1. hash every rank's reads once (parse_count), sum the task sizes, dispatch
    std::vector<u64> bytes(ntasks, 0);
    std::vector<ParseJob> jobs(R);
    auto release_jobs = [&]() { for (auto &j : jobs) parse_release(c, j); };
    for (int r = 0; r < R; ++r) {
        int rc = parse_count(c, in[r].packed, packed_bytes[r], in[r].roff, in[r].rlen, nreads[r], rid_base[r], ntasks, jobs[r]);
        if (rc) { release_jobs(); return rc; }
        for (u32 t = 0; t < ntasks; ++t) bytes[t] += jobs[r].task_tot[3 * t + 1] + jobs[r].task_tot[3 * t] * (ext ? 9 : 1);
    }
    // 1b. heavy-hitter tasks: classify on the global k-mer counts, every rank pre-aggregates its share
    std::vector<u8> is_heavy(ntasks, 0);
    std::vector<std::vector<TaskOut>> hlists(R);
    auto free_hlists = [&]() { for (auto &v : hlists) for (auto &to : v) free_task_out(c, to); };
    bool any_heavy = false;
    if (heavy_enabled(c, NW, R)) {
        std::vector<u64> kg(ntasks, 0); std::vector<int32_t> types(ntasks, 0);
        for (int r = 0; r < R; ++r) for (u32 t = 0; t < ntasks; ++t) kg[t] += jobs[r].task_tot[3 * t + 2];
        plan_classify(kg.data(), (int)ntasks, heavy_ratio(), types.data());
        for (u32 t = 0; t < ntasks; ++t) if (types[t] == 1) { is_heavy[t] = 1; any_heavy = true; }
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
        int rc = parse_place(c, jobs[r], order, st[r], any_heavy ? &is_heavy : nullptr);
        parse_release(c, jobs[r]);
        if (rc) { release_jobs(); return rc; }
        rc = pack_store_bytes(c, st[r], source_from_packed(in[r].packed, packed_bytes[r], st[r].sm_gpos)); if (rc) { release_jobs(); return rc; }
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
        HIPCHK(c, hipStreamSynchronize(c->stream));
        free_hlists();
    }
    auto free_hin = [&]() { for (auto &v : hin) for (auto &hv : v) c->pool.release(hv.d_entries); };
    // 3. the exchange: same plans as the RCCL path (hsk_comm.h), device copies instead of send/recv
    if (overlap_enabled()) {
        // grouped exchange overlapped with the sort, exactly as run_pipeline drives it
        std::vector<GroupFeeder> fd(R);
        std::vector<std::vector<ExchangePlan>> pl_all(R);
        std::vector<std::vector<TaskSegs>> segs(R);
        for (int d = 0; d < R; ++d) { int rc = fd[d].plan(c, R, d, ntasks, owner, order, M, st[d].task_base, segs[d]); if (rc) return rc; pl_all[d] = fd[d].pl; }
        int rc_all = HSK_OK;
        for (int r = 0; r < R && rc_all == HSK_OK; ++r) {
            fd[r].st_all = &st; fd[r].pl_all = &pl_all;
            memset(&outs[r], 0, sizeof(hsk_result));
            ResultPriv *rp = new ResultPriv();
            outs[r].priv = rp; outs[r].nw = NW; outs[r].ntasks = (int32_t)ntasks;
            PhaseTimer pt(c);
            ProcExtra ex; ex.heavy_in = &hin[r];
            rc_all = process_rank<NW>(c, ntasks, owner, r, segs[r], nullptr, BaseSource(), nullptr, nullptr, &outs[r], rp, pt, false, &fd[r], &ex);
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    }
    free_hin();
    return HSK_OK;
}

static int dispatch_pipeline(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                             int64_t rid_base, hsk_result *out)
{
    int rc;
    switch (c->nw) {
    case 1: rc = run_pipeline<1>(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, out); break;
    case 2: rc = run_pipeline<2>(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, out); break;
    default: rc = run_pipeline<3>(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, out); break;
    }
    if (rc != HSK_OK) { (void)hipStreamSynchronize(c->stream); hsk_result_free(c, out); }
    return rc;
}

extern "C" void hsk_result_free(hsk_ctx *c, hsk_result *r)
{
    if (!r) return;
    ResultPriv *rp = (ResultPriv *)r->priv;
    if (rp) {
        for (void *p : rp->host_blocks) (void)hipHostFree(p);
        if (c) for (auto &to : rp->dev_tasks) free_task_out(c, to);
        delete rp;
    }
    memset(r, 0, sizeof *r);
}

extern "C" int hsk_result_device_task(const hsk_result *r, int32_t task, const void **entries, uint64_t *n,
                                      const void **payload_off, const void **pos, const void **rid, uint64_t *npay, uint64_t *payload_base)
{
    if (!r || !r->priv || task < 0 || task >= r->ntasks) return HSK_ERR_INVALID_ARG;
    const ResultPriv *rp = (const ResultPriv *)r->priv;
    if ((size_t)task >= rp->dev_tasks.size()) return HSK_ERR_INVALID_ARG;        // not a KEEP_DEVICE result
    const TaskOut &to = rp->dev_tasks[task];
    if (entries) *entries = to.entries;
    if (n) *n = to.n;
    if (payload_off) *payload_off = to.payoff;
    if (pos) *pos = to.pos;
    if (rid) *rid = to.rid;
    if (npay) *npay = to.npay;
    if (payload_base) *payload_base = to.pay_base;
    return HSK_OK;
}

// Uploads the DnaBuffer description; returns device arrays with nreads+1 offsets.

static int upload_input(hsk_ctx *c, const uint8_t *packed, uint64_t packed_bytes, const uint64_t *off, const uint32_t *len,
                        uint64_t nreads, DevInput &d)
{
    DALLOC(c, d.packed, u8 *, packed_bytes + 64);
    DALLOC(c, d.roff, u64 *, (nreads + 1) * 8);
    DALLOC(c, d.rlen, u32 *, (nreads + 1) * 4);
    if (packed_bytes) HIPCHK(c, hipMemcpyAsync(d.packed, packed, packed_bytes, hipMemcpyHostToDevice, c->stream));
    if (nreads) {
        HIPCHK(c, hipMemcpyAsync(d.roff, off, nreads * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d.rlen, len, nreads * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(d.roff + nreads, &packed_bytes, 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // host buffers (and &packed_bytes) may go away after return
    return HSK_OK;
}
static void free_input(hsk_ctx *c, DevInput &d) { c->pool.release(d.packed); c->pool.release(d.roff); c->pool.release(d.rlen); d = DevInput(); }

static int check_host_index(hsk_ctx *c, uint64_t packed_bytes, const uint64_t *off, const uint32_t *len, uint64_t nreads)
{
    uint64_t prev_end = 0;
    for (uint64_t r = 0; r < nreads; ++r) {
        if (off[r] < prev_end) return fail(c, HSK_ERR_INVALID_ARG, "read %llu overlaps its predecessor", (unsigned long long)r);
        const uint64_t end = off[r] + ((uint64_t)len[r] + 3) / 4;
        if (end > packed_bytes) return fail(c, HSK_ERR_INVALID_ARG, "read %llu extends past the packed buffer", (unsigned long long)r);
        prev_end = end;
    }
    if (nreads && off[0] != 0) return fail(c, HSK_ERR_INVALID_ARG, "read_byte_off[0] must be 0");
    return HSK_OK;
}

extern "C" int hsk_count(hsk_ctx *c, const uint8_t *packed, uint64_t packed_bytes, const uint64_t *off, const uint32_t *len,
                         uint64_t nreads, int64_t rid_base, hsk_result *out)
{
    if (!c || !out || (nreads && (!off || !len)) || (packed_bytes && !packed)) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = check_host_index(c, packed_bytes, off, len, nreads); if (rc) return rc;
    DevInput d;
    rc = upload_input(c, packed, packed_bytes, off, len, nreads, d);
    if (rc == HSK_OK) rc = dispatch_pipeline(c, d.packed, packed_bytes, d.roff, d.rlen, nreads, rid_base, out);
    free_input(c, d);
    return rc;
}

extern "C" int hsk_count_device(hsk_ctx *c, const void *d_packed, uint64_t packed_bytes, const void *d_off, const void *d_len,
                                uint64_t nreads, int64_t rid_base, hsk_result *out)
{
    if (!c || !out || (nreads && (!d_off || !d_len)) || (packed_bytes && !d_packed)) return HSK_ERR_INVALID_ARG;
    if (((uintptr_t)d_packed & 3) != 0) return fail(c, HSK_ERR_INVALID_ARG, "d_packed must be 4-byte aligned");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    // the kernels index roff[r+1]: build the (nreads+1)-entry offset array
    u64 *roff; DALLOC(c, roff, u64 *, (nreads + 1) * 8);
    if (nreads) HIPCHK(c, hipMemcpyAsync(roff, d_off, nreads * 8, hipMemcpyDeviceToDevice, c->stream));
    u64 *stage = (u64 *)((char *)c->pinned + c->pinned_bytes - 256);
    *stage = packed_bytes;
    HIPCHK(c, hipMemcpyAsync(roff + nreads, stage, 8, hipMemcpyHostToDevice, c->stream));
    int rc = dispatch_pipeline(c, (const u8 *)d_packed, packed_bytes, roff, (const u32 *)d_len, nreads, rid_base, out);
    c->pool.release(roff);
    return rc;
}

extern "C" int hsk_count_loopback(hsk_ctx *c, int nranks, const uint8_t *const *packed, const uint64_t *packed_bytes, const uint64_t *const *off,
                                  const uint32_t *const *len, const uint64_t *nreads, hsk_result *outs, int32_t *owner_out, int32_t owner_capacity)
{
    if (!c || nranks < 1 || nranks > 64 || !packed || !packed_bytes || !off || !len || !nreads || !outs) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<DevInput> in(nranks);
    int rc = HSK_OK;
    for (int r = 0; r < nranks && rc == HSK_OK; ++r) {
        rc = check_host_index(c, packed_bytes[r], off[r], len[r], nreads[r]);
        if (rc == HSK_OK) rc = upload_input(c, packed[r], packed_bytes[r], off[r], len[r], nreads[r], in[r]);
    }
    u32 ntasks = 0;
    std::vector<int32_t> owner(HSK_MAX_TASKS, 0);
    if (rc == HSK_OK) {
        switch (c->nw) {
        case 1: rc = run_loopback<1>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        case 2: rc = run_loopback<2>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        default: rc = run_loopback<3>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        }
    }
    if (rc == HSK_OK && owner_out) { if ((u32)owner_capacity < ntasks) rc = HSK_ERR_INVALID_ARG; else memcpy(owner_out, owner.data(), sizeof(int32_t) * ntasks); }
    for (auto &d : in) free_input(c, d);
    if (rc != HSK_OK) { (void)hipStreamSynchronize(c->stream); for (int r = 0; r < nranks; ++r) hsk_result_free(c, &outs[r]); }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// stage entry points
// ------------------------------------------------------------------------------------------------
extern "C" int hsk_stage_destinations(hsk_ctx *c, const uint8_t *packed, uint64_t packed_bytes, const uint64_t *off, const uint32_t *len,
                                      uint64_t nreads, int32_t *dest, uint64_t cap, uint64_t *dest_off)
{
    if (!c || !dest_off || (cap && !dest)) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = check_host_index(c, packed_bytes, off, len, nreads); if (rc) return rc;
    const int K = c->cfg.kmer_size;
    uint64_t total = 0;
    for (uint64_t r = 0; r < nreads; ++r) { dest_off[r] = total; total += len[r] >= (uint32_t)K ? len[r] - K + 1 : 0; }
    dest_off[nreads] = total;
    if (total > cap) return fail(c, HSK_ERR_INVALID_ARG, "dest capacity %llu < %llu", (unsigned long long)cap, (unsigned long long)total);
    if (!nreads || !packed_bytes) return HSK_OK;
    DevInput d; rc = upload_input(c, packed, packed_bytes, off, len, nreads, d); if (rc) return rc;
    const u32 ntasks = c->cfg.ntasks ? (u32)c->cfg.ntasks : 1;
    u32 nblocks; ParseArgs a = make_parse_args(c, d.packed, packed_bytes, d.roff, d.rlen, nreads, 0, ntasks, &nblocks);
    int32_t *d_dump; DALLOC(c, d_dump, int32_t *, packed_bytes * 4 * 4);
    a.dump_dest = d_dump;
    hipLaunchKernelGGL((parse_kernel<PARSE_DUMP, false>), dim3(nblocks), dim3(PARSE_THREADS), 64, c->stream, a);
    std::vector<int32_t> h(packed_bytes * 4);
    HIPCHK(c, hipMemcpyAsync(h.data(), d_dump, packed_bytes * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint64_t r = 0; r < nreads; ++r) {
        const uint64_t nk = dest_off[r + 1] - dest_off[r];
        for (uint64_t i = 0; i < nk; ++i) dest[dest_off[r] + i] = h[off[r] * 4 + i];
    }
    c->pool.release(d_dump); free_input(c, d);
    return HSK_OK;
}

template <int NW>
static int stage_task_kmers_impl(hsk_ctx *c, const DevInput &d, uint64_t packed_bytes, uint64_t nreads, int64_t rid_base, int32_t task,
                                 uint64_t *keys, uint32_t *pos, int32_t *rid, uint64_t cap, uint64_t *n)
{
    const bool ext = c->cfg.extension != 0;
    const u32 ntasks = c->cfg.ntasks ? (u32)c->cfg.ntasks : 1;
    if (task < 0 || (u32)task >= ntasks) return HSK_ERR_INVALID_ARG;
    std::vector<u32> order(ntasks); for (u32 t = 0; t < ntasks; ++t) order[t] = t;
    SupermerStore st;
    int rc = parse_phase(c, d.packed, packed_bytes, d.roff, d.rlen, nreads, rid_base, ntasks, order, st); if (rc) return rc;
    TaskSegs ts;
    if (st.task_tot[3 * task]) {
        ExpSeg s; s.sup_off = st.task_base[3 * task]; s.n_sup = st.task_tot[3 * task]; s.byte_off = st.task_base[3 * task + 1]; s.kmer_off = 0; s.tile_start = 0;
        ts.segs.push_back(s); ts.nkmers = st.task_tot[3 * task + 2];
    }
    finalize_segs(ts);
    *n = ts.nkmers;
    if (ts.nkmers > cap) { free_store(c, st); return fail(c, HSK_ERR_INVALID_ARG, "capacity %llu < %llu", (unsigned long long)cap, (unsigned long long)ts.nkmers); }
    if (ts.nkmers) {
        u64 *dk, *dv = nullptr;
        DALLOC(c, dk, u64 *, ts.nkmers * NW * 8 + 64);
        if (ext) DALLOC(c, dv, u64 *, ts.nkmers * 8 + 64);
        rc = expand_task<NW>(c, ts, st.sm_len, source_from_packed(d.packed, packed_bytes, st.sm_gpos), st.sm_pos, st.sm_rid, dk, dv);
        if (rc == HSK_OK) {
            HIPCHK(c, hipMemcpyAsync(keys, dk, ts.nkmers * NW * 8, hipMemcpyDeviceToHost, c->stream));
            std::vector<u64> hv;
            if (ext) { hv.resize(ts.nkmers); HIPCHK(c, hipMemcpyAsync(hv.data(), dv, ts.nkmers * 8, hipMemcpyDeviceToHost, c->stream)); }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (ext) for (u64 i = 0; i < ts.nkmers; ++i) { if (pos) pos[i] = (uint32_t)hv[i]; if (rid) rid[i] = (int32_t)(hv[i] >> 32); }
        }
        c->pool.release(dk); c->pool.release(dv);
    }
    free_store(c, st);
    return rc;
}

extern "C" int hsk_stage_task_kmers(hsk_ctx *c, const uint8_t *packed, uint64_t packed_bytes, const uint64_t *off, const uint32_t *len,
                                    uint64_t nreads, int64_t rid_base, int32_t task, uint64_t *keys, uint32_t *pos, int32_t *rid,
                                    uint64_t cap, uint64_t *n)
{
    if (!c || !n || (cap && !keys)) return HSK_ERR_INVALID_ARG;
    *n = 0;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = check_host_index(c, packed_bytes, off, len, nreads); if (rc) return rc;
    if (!nreads || !packed_bytes) return HSK_OK;
    DevInput d; rc = upload_input(c, packed, packed_bytes, off, len, nreads, d); if (rc) return rc;
    switch (c->nw) {
    case 1: rc = stage_task_kmers_impl<1>(c, d, packed_bytes, nreads, rid_base, task, keys, pos, rid, cap, n); break;
    case 2: rc = stage_task_kmers_impl<2>(c, d, packed_bytes, nreads, rid_base, task, keys, pos, rid, cap, n); break;
    default: rc = stage_task_kmers_impl<3>(c, d, packed_bytes, nreads, rid_base, task, keys, pos, rid, cap, n); break;
    }
    free_input(c, d);
    return rc;
}

template <int NW>
static int stage_sort_impl(hsk_ctx *c, uint64_t *keys, uint64_t *vals, uint64_t n)
{
    u64 *ka, *kb, *va = nullptr, *vb = nullptr;
    DALLOC(c, ka, u64 *, n * NW * 8 + 64); DALLOC(c, kb, u64 *, n * NW * 8 + 64);
    if (vals) { DALLOC(c, va, u64 *, n * 8 + 64); DALLOC(c, vb, u64 *, n * 8 + 64); }
    HIPCHK(c, hipMemcpyAsync(ka, keys, n * NW * 8, hipMemcpyHostToDevice, c->stream));
    if (vals) HIPCHK(c, hipMemcpyAsync(va, vals, n * 8, hipMemcpyHostToDevice, c->stream));
    SortScratch sc; int rc = alloc_sort_scratch(c, sc); if (rc) return rc;
    u64 *sk, *sv;
    // all 64 bits of every word take part (K = 32*NW would be the natural name; 32*NW-... use full words)
    rc = sort_task_device<NW>(c, ka, kb, va, vb, n, 32 * NW, sc, &sk, &sv);
    if (rc == HSK_OK) rc = check_device_error(c);
    if (rc == HSK_OK) {
        HIPCHK(c, hipMemcpyAsync(keys, sk, n * NW * 8, hipMemcpyDeviceToHost, c->stream));
        if (vals) HIPCHK(c, hipMemcpyAsync(vals, sv, n * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    free_sort_scratch(c, sc);
    c->pool.release(ka); c->pool.release(kb); c->pool.release(va); c->pool.release(vb);
    return rc;
}

extern "C" int hsk_stage_sort(hsk_ctx *c, uint64_t *keys, uint64_t *vals, uint64_t n, int32_t nw)
{
    if (!c || (n && !keys) || nw < 1 || nw > 3) return HSK_ERR_INVALID_ARG;
    if (n == 0) return HSK_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    switch (nw) {
    case 1: return stage_sort_impl<1>(c, keys, vals, n);
    case 2: return stage_sort_impl<2>(c, keys, vals, n);
    default: return stage_sort_impl<3>(c, keys, vals, n);
    }
}

template <int NW>
static int stage_count_impl(hsk_ctx *c, const uint64_t *keys, uint64_t n, uint64_t *out_entries, uint64_t cap, uint64_t *n_out)
{
    u64 *dk; DALLOC(c, dk, u64 *, n * NW * 8 + 64);
    HIPCHK(c, hipMemcpyAsync(dk, keys, n * NW * 8, hipMemcpyHostToDevice, c->stream));
    const u32 histo_len = (u32)std::min<int64_t>((int64_t)c->cfg.upper_freq + 1, 65536);    // (U <= 65535 except in the unfiltered pre-aggregation)
    u64 *d_histo; DALLOC(c, d_histo, u64 *, (size_t)histo_len * 8);
    HIPCHK(c, hipMemsetAsync(d_histo, 0, (size_t)histo_len * 8, c->stream));
    TaskOut to;
    int rc = count_task_device<NW>(c, dk, nullptr, n, 0, d_histo, histo_len, to);
    if (rc == HSK_OK) {
        *n_out = to.n;
        if (to.n > cap) rc = fail(c, HSK_ERR_INVALID_ARG, "capacity %llu < %llu", (unsigned long long)cap, (unsigned long long)to.n);
        else if (to.n) HIPCHK(c, hipMemcpyAsync(out_entries, to.entries, to.n * (NW + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    free_task_out(c, to);
    c->pool.release(dk); c->pool.release(d_histo);
    return rc;
}

extern "C" int hsk_stage_count_sorted(hsk_ctx *c, const uint64_t *keys, uint64_t n, int32_t nw, uint64_t *out_entries, uint64_t cap, uint64_t *n_out)
{
    if (!c || !n_out || (n && !keys) || nw < 1 || nw > 3) return HSK_ERR_INVALID_ARG;
    *n_out = 0;
    if (n == 0) return HSK_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    switch (nw) {
    case 1: return stage_count_impl<1>(c, keys, n, out_entries, cap, n_out);
    case 2: return stage_count_impl<2>(c, keys, n, out_entries, cap, n_out);
    default: return stage_count_impl<3>(c, keys, n, out_entries, cap, n_out);
    }
}

// ------------------------------------------------------------------------------------------------
// host planning (pure CPU)
// ------------------------------------------------------------------------------------------------
extern "C" int hsk_plan_tot_tasks(int omp_max_threads, int thread_per_worker, int avg_task_per_worker, int nprocs)
{
    if (thread_per_worker < 1 || nprocs < 1) return -1;
    return plan_tot_tasks(omp_max_threads, thread_per_worker, avg_task_per_worker, nprocs);
}
extern "C" int hsk_plan_classify(const uint64_t *task_kmers, int ntasks, double ratio, int32_t *types)
{
    if (!task_kmers || !types || ntasks < 1) return HSK_ERR_INVALID_ARG;
    plan_classify(task_kmers, ntasks, ratio, types);
    return HSK_OK;
}
extern "C" int hsk_plan_dispatch(const uint64_t *task_bytes, int ntasks, int nprocs, int plain, double upper_coe, double step, int32_t *owner)
{
    if (!task_bytes || !owner || ntasks < 1 || nprocs < 1) return HSK_ERR_INVALID_ARG;
    int rc = plan_dispatch(task_bytes, ntasks, nprocs, plain != 0, upper_coe, step, owner);
    return rc == 0 ? HSK_OK : (rc == -1 ? HSK_ERR_DISPATCH : HSK_ERR_INVALID_ARG);
}
extern "C" int hsk_plan_partition_reads(const uint64_t *read_len, uint64_t nreads, int nprocs, uint64_t *counts)
{
    if (!counts || nprocs < 1 || (nreads && !read_len)) return HSK_ERR_INVALID_ARG;
    return plan_partition_reads(read_len, nreads, nprocs, counts) == 0 ? HSK_OK : HSK_ERR_INVALID_ARG;
}

extern "C" int hsk_plan_exchange(int nranks, int rank, int ntasks, const int32_t *owner, const uint64_t *size_matrix,
                                 uint64_t *send_recv, uint64_t *segs_out)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || ntasks < 1 || !owner || !size_matrix || !send_recv || !segs_out) return HSK_ERR_INVALID_ARG;
    std::vector<int32_t> own(owner, owner + ntasks);
    for (int t = 0; t < ntasks; ++t) if (own[t] < 0 || own[t] >= nranks) return HSK_ERR_INVALID_ARG;
    std::vector<u32> order(ntasks);
    for (int t = 0; t < ntasks; ++t) order[t] = (u32)t;
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return own[x] < own[y]; });
    std::vector<u64> M(size_matrix, size_matrix + (size_t)nranks * ntasks * 3);
    // storage bases of this rank's own supermers: exclusive prefix over the storage order
    std::vector<u64> base((size_t)ntasks * 3, 0);
    u64 s0 = 0, b0 = 0, k0 = 0;
    for (int i = 0; i < ntasks; ++i) {
        const u32 t = order[i];
        base[3 * t] = s0; base[3 * t + 1] = b0; base[3 * t + 2] = k0;
        const u64 *m = &M[((size_t)rank * ntasks + t) * 3];
        s0 += m[0]; b0 += m[1]; k0 += m[2];
    }
    ExchangePlan pl; std::vector<TaskSegs> segs;
    plan_exchange(nranks, rank, (u32)ntasks, own, order, M, base, pl, segs);
    for (int q = 0; q < nranks; ++q) {
        u64 *o = send_recv + (size_t)q * 8;
        o[0] = pl.send_sup[q]; o[1] = pl.send_bytes[q]; o[2] = pl.send_sup_off[q]; o[3] = pl.send_byte_off[q];
        o[4] = pl.recv_sup[q]; o[5] = pl.recv_bytes[q]; o[6] = pl.recv_sup_off[q]; o[7] = pl.recv_byte_off[q];
    }
    memset(segs_out, 0, sizeof(u64) * (size_t)ntasks * nranks * 4);
    for (int t = 0; t < ntasks; ++t) {
        if (own[t] != rank) continue;
        // plan_exchange drops empty segments; re-derive the per-source rows so that the table is dense
        u64 koff = 0;
        size_t si = 0;
        for (int p = 0; p < nranks; ++p) {
            const u64 *m = &M[((size_t)p * ntasks + t) * 3];
            u64 *o = segs_out + ((size_t)t * nranks + p) * 4;
            if (m[0]) { const ExpSeg &sg = segs[t].segs[si++]; o[0] = sg.sup_off; o[1] = sg.n_sup; o[2] = sg.byte_off; o[3] = sg.kmer_off; }
            else { o[3] = koff; }
            koff += m[2];
        }
    }
    return HSK_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU
// ------------------------------------------------------------------------------------------------
extern "C" int hsk_comm_get_unique_id(void *id128)
{
    if (!id128) return HSK_ERR_INVALID_ARG;
    return Comm::get_unique_id(id128) == 0 ? HSK_OK : HSK_ERR_COMM;
}
extern "C" int hsk_comm_init(hsk_ctx *c, int nranks, int rank, const void *id128)
{
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = c->comm.init(nranks, rank, id128);
    if (rc) return fail(c, HSK_ERR_COMM, "RCCL init failed: %s", c->comm.last_error.c_str());
    return HSK_OK;
}
// One-rank communicator on this ctx's GPU: every RCCL entry point the exchange uses (unique id, init, all-reduce
// sum/max of u64, grouped send/recv of bytes to self on the second stream, destroy) with checked results.  This is
// how the RCCL binding (dlopen'ed symbols, enum values, by-value unique id) is exercised on a single-GPU box.
extern "C" int hsk_comm_selftest(hsk_ctx *c)
{
    if (!c) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    char id[HSK_UNIQUE_ID_BYTES];
    if (Comm::get_unique_id(id) != 0) return fail(c, HSK_ERR_COMM, "ncclGetUniqueId failed (librccl not loadable?)");
    Comm cm;
    int rc = cm.init(1, 0, id, true);
    if (rc) return fail(c, HSK_ERR_COMM, "RCCL init failed: %s", cm.last_error.c_str());
    const size_t n = 1 << 20;
    u64 *d_a; u8 *d_src, *d_dst;
    DALLOC(c, d_a, u64 *, 4096 * 8); DALLOC(c, d_src, u8 *, n); DALLOC(c, d_dst, u8 *, n);
    std::vector<u64> h(4096), h2(4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = splitmix64(i);
    std::vector<u8> hs(n), hd(n, 0);
    for (size_t i = 0; i < n; ++i) hs[i] = (u8)(splitmix64(i) >> 13);
    int out = HSK_OK;
    do {
        if (hipMemcpyAsync(d_a, h.data(), h.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_src, hs.data(), n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemsetAsync(d_dst, 0, n, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { out = fail(c, HSK_ERR_HIP, "selftest upload"); break; }
        if ((rc = cm.check(cm.api->AllReduce(d_a, d_a, h.size(), RCCL_UINT64, RCCL_SUM, cm.comm, c->stream), "ncclAllReduce(sum)")) ||
            (rc = cm.check(cm.api->AllReduce(d_a, d_a, h.size(), RCCL_UINT64, RCCL_MAX, cm.comm, c->stream), "ncclAllReduce(max)"))) { out = fail(c, HSK_ERR_COMM, "%s", cm.last_error.c_str()); break; }
        if (hipMemcpyAsync(h2.data(), d_a, h.size() * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { out = fail(c, HSK_ERR_HIP, "selftest download"); break; }
        if (h2 != h) { out = fail(c, HSK_ERR_COMM, "one-rank all-reduce changed the data"); break; }
        // two messages to self inside one group, on the second stream (as post_exchange does per peer and array)
        hipStream_t s = c->comm_stream;
        if ((rc = cm.check(cm.api->GroupStart(), "ncclGroupStart")) ||
            (rc = cm.check(cm.api->Send(d_src, n / 2, RCCL_UINT8, 0, cm.comm, s), "ncclSend")) ||
            (rc = cm.check(cm.api->Send(d_src + n / 2, n - n / 2, RCCL_UINT8, 0, cm.comm, s), "ncclSend")) ||
            (rc = cm.check(cm.api->Recv(d_dst, n / 2, RCCL_UINT8, 0, cm.comm, s), "ncclRecv")) ||
            (rc = cm.check(cm.api->Recv(d_dst + n / 2, n - n / 2, RCCL_UINT8, 0, cm.comm, s), "ncclRecv")) ||
            (rc = cm.check(cm.api->GroupEnd(), "ncclGroupEnd"))) { out = fail(c, HSK_ERR_COMM, "%s", cm.last_error.c_str()); break; }
        if (hipMemcpyAsync(hd.data(), d_dst, n, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { out = fail(c, HSK_ERR_HIP, "selftest download"); break; }
        if (hd != hs) { out = fail(c, HSK_ERR_COMM, "grouped send/recv to self delivered different bytes"); break; }
    } while (0);
    cm.destroy();
    c->pool.release(d_a); c->pool.release(d_src); c->pool.release(d_dst);
    return out;
}

extern "C" int hsk_comm_destroy(hsk_ctx *c)
{
    if (!c) return HSK_ERR_INVALID_ARG;
    c->comm.destroy();
    return HSK_OK;
}

// ------------------------------------------------------------------------------------------------
// synthetic reads in HBM
// ------------------------------------------------------------------------------------------------
extern "C" int hsk_synth_reads(hsk_ctx *c, uint64_t genome_len, uint32_t read_len, uint64_t nreads, uint64_t seed, uint64_t first_read,
                               void **d_packed, uint64_t *packed_bytes, void **d_off, void **d_len)
{
    if (!c || !d_packed || !packed_bytes || !d_off || !d_len || read_len == 0 || genome_len < read_len) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const u64 nwords = (genome_len + 31) / 32;
    const u32 nb = (read_len + 3) / 4;
    const u64 bytes = nreads * nb;
    u64 *gw; u8 *pk; u64 *roff; u32 *rlen;
    DALLOC(c, gw, u64 *, nwords * 8);
    DALLOC(c, pk, u8 *, bytes + 64);
    DALLOC(c, roff, u64 *, (nreads + 1) * 8);
    DALLOC(c, rlen, u32 *, (nreads + 1) * 4);
    hipLaunchKernelGGL(synth_genome_kernel, dim3((u32)((nwords + 255) / 256)), dim3(256), 0, c->stream, gw, nwords, seed);
    const u64 seed2 = splitmix64(seed ^ 0xabcdef12345ULL);
    if (bytes) hipLaunchKernelGGL(synth_reads_kernel, dim3((u32)((bytes + 255) / 256)), dim3(256), 0, c->stream, gw, genome_len, read_len, nreads, seed2 + first_read, pk);
    hipLaunchKernelGGL(synth_index_kernel, dim3((u32)((nreads + 256) / 256)), dim3(256), 0, c->stream, roff, rlen, nreads, read_len);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pool.release(gw);
    *d_packed = pk; *packed_bytes = bytes; *d_off = roff; *d_len = rlen;
    return HSK_OK;
}

extern "C" int hsk_pack_fasta(hsk_ctx *c, const char *text, uint64_t text_bytes, const uint64_t *rec_pos, const uint32_t *rec_len,
                              const uint32_t *line_bases, const uint32_t *line_width, uint64_t nrec,
                              void **d_packed, uint64_t *packed_bytes, void **d_off, void **d_len)
{
    if (!c || !d_packed || !packed_bytes || !d_off || !d_len || (nrec && (!text || !rec_pos || !rec_len || !line_bases || !line_width))) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    *d_packed = nullptr; *d_off = nullptr; *d_len = nullptr; *packed_bytes = 0;
    uint64_t total = 0;
    for (uint64_t r = 0; r < nrec; ++r) {
        const uint64_t nl = line_bases[r] ? ((uint64_t)rec_len[r] + line_bases[r] - 1) / line_bases[r] : 0;
        const uint64_t last = rec_len[r] ? rec_pos[r] + (line_bases[r] ? (nl - 1) * (uint64_t)line_width[r] + ((uint64_t)rec_len[r] - (nl - 1) * line_bases[r]) : rec_len[r]) : rec_pos[r];
        if (last > text_bytes) return fail(c, HSK_ERR_INVALID_ARG, "record %llu extends past the text", (unsigned long long)r);
        if (line_bases[r] && line_width[r] < line_bases[r]) return fail(c, HSK_ERR_INVALID_ARG, "record %llu: line width < bases per line", (unsigned long long)r);
        total += ((uint64_t)rec_len[r] + 3) / 4;
    }
    u8 *d_text, *pk; u64 *d_pos, *roff; u32 *rlen, *d_lb, *d_lw;
    DALLOC(c, pk, u8 *, total + 64);
    DALLOC(c, roff, u64 *, (nrec + 1) * 8);
    DALLOC(c, rlen, u32 *, (nrec + 1) * 4);
    if (nrec) {
        DALLOC(c, d_text, u8 *, text_bytes + 64);
        DALLOC(c, d_pos, u64 *, nrec * 8); DALLOC(c, d_lb, u32 *, nrec * 4); DALLOC(c, d_lw, u32 *, nrec * 4);
        HIPCHK(c, hipMemcpyAsync(d_text, text, text_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_pos, rec_pos, nrec * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(rlen, rec_len, nrec * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_lb, line_bases, nrec * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_lw, line_width, nrec * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(pack_fasta_offsets_kernel, dim3(1), dim3(256), 0, c->stream, rlen, nrec, roff);
        if (total) hipLaunchKernelGGL(pack_fasta_kernel, dim3((u32)std::min<u64>((total + 255) / 256, 1u << 20)), dim3(256), 0, c->stream,
                                      d_text, text_bytes, d_pos, rlen, d_lb, d_lw, roff, nrec, total, pk);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->pool.release(d_text); c->pool.release(d_pos); c->pool.release(d_lb); c->pool.release(d_lw);
    }
    *d_packed = pk; *packed_bytes = total; *d_off = roff; *d_len = rlen;
    return HSK_OK;
}

extern "C" int hsk_synth_free(hsk_ctx *c, void *d_packed, void *d_off, void *d_len)
{
    if (!c) return HSK_ERR_INVALID_ARG;
    c->pool.release(d_packed); c->pool.release(d_off); c->pool.release(d_len);
    return HSK_OK;
}

extern "C" int hsk_memcpy_d2h(hsk_ctx *c, void *dst, const void *d_src, uint64_t bytes)
{
    if (!c || !dst || !d_src) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HSK_OK;
}

// diagnostic build only: phase clock sums of the onesweep kernel (zeros in the product build)
extern "C" int hsk_debug_diag(unsigned long long *out, int n, int reset)
{
    if (!out || n < 1 || n > 32) return HSK_ERR_INVALID_ARG;
    memset(out, 0, sizeof(unsigned long long) * n);
#ifdef HSK_DIAG
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hsk::g_diag), sizeof(unsigned long long) * n) != hipSuccess) return HSK_ERR_HIP;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hsk::g_diag), z, sizeof z) != hipSuccess) return HSK_ERR_HIP; }
#else
    (void)reset;
#endif
    return HSK_OK;
}
