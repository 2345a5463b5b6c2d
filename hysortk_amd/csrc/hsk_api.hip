// hsk_api.hip -- host orchestration + C ABI of libhsk.so (see include/hsk.h).
//
// One translation unit, compiled with `hipcc --offload-arch=gfx950`.  The pipeline behind
// hsk_count() / hsk_count_device() (host side in the hsk_host_*.h files, kernels in the files named on the right):
//
//   scan -> totals -> [all-reduce, dispatch] -> place      (hsk_parse.h)   reads -> per-task supermers
//   [heavy-hitter lists, RCCL all-to-all of supermers]     (hsk_comm.h)    multi-GPU only
//   per batch of eight owned tasks, ascending id:
//     expand (tile sums, scan, extract + digit histograms) (hsk_expand.h)  supermers -> canonical k-mers
//     scatter passes                                       (hsk_sort.h)    LSD radix on the key prefix (or all bits)
//     aggregation / merge-count + [L,U] filter             (hsk_agg.h, hsk_count.h, hsk_finish.h, hsk_heavy.h)
//   result assembly (pinned host memory, or left in HBM)
//
// There is no CPU fallback anywhere in this file: without a gfx950 device hsk_init() fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hsk.h"
#include "hsk_device.h"
#include "hsk_parse.h"
#include "hsk_expand.h"
#include "hsk_sort.h"
#include "hsk_scatter.h"
#include "hsk_count.h"
#include "hsk_finish.h"
#include "hsk_agg.h"
#include "hsk_combine.h"
#include "hsk_heavy.h"
#include "hsk_estimate.h"
#include "hsk_synth.h"
#include "hsk_plan.h"
#include "hsk_comm.h"

using namespace hsk;

#include "hsk_pool.h"
#include "hsk_host_ctx.h"

// ------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" int hsk_abi_version(void) { return HSK_ABI_VERSION; }

// pinned (page-locked, device-visible) host memory for the caller's DnaBuffer: hsk_count() then reads it in place
extern "C" void *hsk_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void hsk_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int hsk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

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
    cfg->dispatch_upper_coe = 1.5; cfg->dispatch_step = 0.05; cfg->radix_bits = 8; cfg->flags = 0; cfg->unbalanced_ratio = 2.3;
}

static int validate_cfg(const hsk_config *cfg)
{
    const int K = cfg->kmer_size, M = cfg->minimizer_size;
    if (!(2 < K && K < 96) || (K % 32) == 0) return 0;            // compiletime.h:10; K%32==0 is UB in the reference
    if (!(0 < M && M < K) || (M % 32) == 0) return 0;               // Makefile:50-52; M % 32 == 0 is the same undefined shift in Mmer::GetExtension (supermer.hpp:265)
    if (!(0 < cfg->lower_freq && cfg->lower_freq <= cfg->upper_freq && cfg->upper_freq <= 65535)) return 0;  // compiletime.h:21
    if (cfg->extension != 0 && cfg->extension != 1) return 0;
    if (cfg->ntasks < 0 || cfg->ntasks > HSK_MAX_TASKS) return 0;
    if (cfg->radix_bits != 0 && (cfg->radix_bits < 4 || cfg->radix_bits > 8)) return 0;
    return 1;
}

// HSK_BACKTRACE=1 (diagnostic): print the native stack when the process is aborted (a runtime that gives up on a queue
// error calls abort() without saying where from)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void backtrace_on_abort(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

extern "C" int hsk_init(const hsk_config *cfg, hsk_ctx **out)
{
    if (getenv("HSK_BACKTRACE") && atoi(getenv("HSK_BACKTRACE")) != 0) { signal(SIGABRT, backtrace_on_abort); signal(SIGSEGV, backtrace_on_abort); }
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
    if (!(c->cfg.unbalanced_ratio > 0)) c->cfg.unbalanced_ratio = 2.3;
    c->nw = (cfg->kmer_size + 31) / 32;
    c->tune.parse(cfg->tuning);                                   // the client's string first, then the environment's (a name already set stays)
    c->tune.parse(getenv("HSK_TUNING"));
    c->cfg.tuning = nullptr;                                      // (the caller's string is not kept)
    g_tune = &c->tune;
    memset(&c->stats, 0, sizeof c->stats);
    c->pool.be_malloc = pool_hip_malloc; c->pool.be_free = pool_hip_free;
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return HSK_ERR_HIP; }
    c->pinned_bytes = 1 << 20;
    if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault) != hipSuccess) { delete c; return HSK_ERR_OOM; }
    c->comm.stage = (char *)c->pinned + (512u << 10); c->comm.stage_bytes = 256u << 10;     // second half of the staging area, 256 KB
    c->d_err = (u32 *)c->pool.alloc(256);
    if (!c->d_err) { delete c; return HSK_ERR_OOM; }
    (void)hipMemsetAsync(c->d_err, 0, 256, c->stream);
    {   // XCD census: words 16..31 of the error block are scratch here
        u32 *d_cnt = c->d_err + 16, h_cnt[16] = {0};
        hipLaunchKernelGGL(xcc_census_kernel, dim3(4096), dim3(64), 0, c->stream, d_cnt);
        if (hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            hsk_destroy(c); return HSK_ERR_HIP;
        }
        (void)hipMemsetAsync(d_cnt, 0, sizeof h_cnt, c->stream);
        int seen = 0; u32 lo = ~0u;
        for (int i = 0; i < 8; ++i) { if (h_cnt[i]) ++seen; lo = std::min(lo, h_cnt[i]); }
        for (int i = 8; i < 16; ++i) if (h_cnt[i]) seen = -100;
        // every XCD must get a fair share of a round-robin launch (4096 workgroups: 512 each)
        c->xcd_batch_ok = seen == 8 && lo >= 256;
        if (c->tune.get("force_no_xcd", 0)) c->xcd_batch_ok = false;      // test hook
    }
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
    c->hpool.destroy();
    if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
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
            else if (p.kind == 2) { if (p.keys) c->stats.agg_launches++; c->stats.agg_bytes += p.bytes; c->stats.agg_ms += f; }      // (a rung with no listed bins is no launch of work)
            else if (p.kind == 3) { c->stats.scan_launches++; c->stats.scan_bytes += p.bytes; c->stats.scan_ms += f; }
            else if (p.kind == 4) { c->stats.place_launches++; c->stats.place_supermers += p.keys; c->stats.place_ms += f; }
            else if (p.kind == 5) { c->stats.h2d_ms += f; }
            else if (p.kind == 6) { c->stats.d2h_ms += f; }
            else if (p.kind == 7) { c->stats.combine_launches++; c->stats.combine_kmers += p.keys; c->stats.combine_ms += f; }
            else if (p.kind == 8) { c->stats.bucket_launches += 2; c->stats.bucket_items += p.keys; c->stats.bucket_ms += f; }
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

#include "hsk_host_parse.h"
#include "hsk_host_expand.h"
#include "hsk_host_sort.h"
#include "hsk_host_scatter.h"
#include "hsk_host_combine.h"
#include "hsk_host_finish.h"
#include "hsk_host_pipeline.h"

// ------------------------------------------------------------------------------------------------
// The plan is chosen INSIDE the call (hysortk::kmer_count() is called once per process, reference src/hysortk.cpp:36-96: there is
// no "next call" that could profit from what this one learned).  Before anything is parsed, a sketch of the input is counted
// (hsk_estimate.h: the reads inside the first 1/64 of the packed buffer (4 - 64 MB), 1/32 of their canonical k-mers by hash, a global table):
// n1, n2, n3 = chosen k-mers seen once, twice, three times in the sample.  Two components explain them: genomic k-mers, Poisson with
// mean lambda_s copies inside the sample (lambda_s = 3 n3 / n2, their number G = 2 n2 exp(lambda_s) / lambda_s^2 -- doubletons and
// tripletons are nearly free of sequencing errors), and k-mers that occur once whatever the depth (errors, a uniform input):
// E_s = n1 - G lambda_s exp(-lambda_s).  In the whole input (1 / f times the sample) the first kind is seen at least once with
// probability 1 - exp(-lambda_s / f), the second kind grows with the input:
//     distinct per k-mer = (f G (1 - exp(-lambda_s / f)) + E_s) / N_s.
// Measured on the 10 Gbp workload: error-free 32x reads 0.039 (the combining extraction then writes one pair per 25.6 k-mers), 0.3 %
// substitution errors 0.13, 1 % 0.31, uniform reads 1.0.  Reads in genome order (a sorted alignment turned back into reads) make the
// prefix deeper than the model thinks and the first term smaller than it is; the second term, which is what moves the decision for
// deep data, is unaffected.  Cost: two small kernels and one wait.  tuning "plan_sample=0" turns the estimate off (the context's memory of
// earlier calls decides, as in rounds 2-3).
// ------------------------------------------------------------------------------------------------
static int estimate_plan(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads, int nranks)
{
    c->est = PlanEstimate();
    const bool enabled = tune("plan_sample", 1) != 0;
    constexpr u64 MIN_INPUT = 32ULL << 20, MIN_SAMPLE = 4ULL << 20, MAX_SAMPLE = 64ULL << 20;
    // who would use it: one-word keys without payload on one GPU (combining extraction or not, first table, aggregation or not)
    if (!enabled || c->cfg.kmer_size >= 64 || packed_bytes < (u64)tune("plan_min_input", (long long)MIN_INPUT) || nreads < 4096) return HSK_OK;       // (several ranks: every rank sketches its own reads, run_pipeline makes them agree)
    // (payloads, three-word keys, pinned plans: nothing to choose -- the sketch still says which k-mers are certain to be dropped, `valid` stays false)
    const bool plan_wanted = c->nw <= 2 && !c->cfg.extension && !(c->cfg.flags & (HSK_FLAG_NO_AGGREGATION | HSK_FLAG_FULL_SORT));
    if (!plan_wanted && c->cfg.kmer_size > 57) return HSK_OK;
    const auto t0 = std::chrono::steady_clock::now();
    u64 want = std::min<u64>(std::max<u64>(packed_bytes / 64, MIN_SAMPLE), MAX_SAMPLE);      // (10 Gbp: 40 MB of reads, ~4 M chosen k-mer instances: 1.2 ms of kernels)
    u64 lost = 0, n1 = 0, n2 = 0, n3 = 0, ds = 0, ns = 0, s_bytes = 0;
    unsigned long long *h_out = (unsigned long long *)((char *)c->pinned + c->pinned_bytes - 512);
    for (int round = 0; round < 2; ++round) {
        const u64 exp_ins = want * 4 / (1ULL << EST_SELECT_BITS) + 1024;
        u64 cap = 1ULL << 16; while (cap < exp_ins * 4) cap <<= 1;
        unsigned long long *d_keys, *d_out; u32 *d_cnts;
        DALLOC(c, d_keys, unsigned long long *, cap * 8); DALLOC(c, d_cnts, u32 *, cap * 4); DALLOC(c, d_out, unsigned long long *, 256);
        auto release = [&]() { c->pool.release(d_keys); c->pool.release(d_cnts); c->pool.release(d_out); };
        HIPCHK(c, hipMemsetAsync(d_keys, 0, cap * 8, c->stream)); HIPCHK(c, hipMemsetAsync(d_cnts, 0, cap * 4, c->stream)); HIPCHK(c, hipMemsetAsync(d_out, 0, 256, c->stream));
        // the reads that lie completely inside the first `want` bytes (found on the device: the read index of a device-resident input is not on the host)
        hipLaunchKernelGGL(prefix_reads_kernel, dim3(1), dim3(64), 0, c->stream, d_roff, nreads, want, (u64 *)(d_out + 8));
        HIPCHK(c, hipMemcpyAsync(h_out + 8, d_out + 8, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hsk_sync(c, c->stream));
        const u64 s_reads = h_out[8]; s_bytes = h_out[9];
        if (s_reads < 2048 || s_bytes < std::min<u64>(MIN_SAMPLE / 2, (u64)tune("plan_min_input", (long long)MIN_INPUT)) || s_bytes > packed_bytes || s_bytes > want + (1u << 20)) { release(); return HSK_OK; }      // (very long reads, a strange index: no estimate, the context's memory decides)
        // host input: the sample's bytes first (the main run copies them again with its first slab)
        if (c->h2d_src || c->zc_src) HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed), c->h2d_src ? c->h2d_src : c->zc_src, s_bytes, hipMemcpyDefault, c->stream));
        EstimateArgs ea; memset(&ea, 0, sizeof ea);
        ea.packed = d_packed; ea.roff = d_roff; ea.rlen = d_rlen; ea.nreads = s_reads; ea.positions = s_bytes * 4; ea.k = c->cfg.kmer_size;
        ea.keys = d_keys; ea.cnts = d_cnts; ea.cap_mask = cap - 1; ea.out = d_out;
        const u64 nthreads = (ea.positions + EST_SPAN - 1) / EST_SPAN;
        hipLaunchKernelGGL(estimate_insert_kernel, dim3((u32)((nthreads + EST_THREADS - 1) / EST_THREADS)), dim3(EST_THREADS), 0, c->stream, ea);
        hipLaunchKernelGGL(estimate_hist_kernel, dim3((u32)std::min<u64>(cap / EST_THREADS, 4096)), dim3(EST_THREADS), 0, c->stream, ea);
        HIPCHK(c, hipMemcpyAsync(h_out, d_out, 64, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hsk_sync(c, c->stream));
        release();                                                          // (stream-ordered reuse: the kernels above are done)
        lost = h_out[0]; n1 = h_out[1]; n2 = h_out[2]; n3 = h_out[3]; ds = h_out[4]; ns = h_out[5];
        c->est.homo_at = h_out[6]; c->est.homo_cg = h_out[7];
        // shallow data (coverage below ~3): too few tripletons in 1/64 of the reads to tell the depth -- once more on sixteen times as many
        if (round == 0 && !lost && n2 >= 16 && n3 < 256 && want * 16 <= packed_bytes / 2 && want * 16 <= (512ULL << 20)) { want *= 16; continue; }
        break;
    }
    if (lost || ns < (1u << 14) || !plan_wanted) return HSK_OK;            // (no estimate)
    PlanEstimate &e = c->est;
    // several ranks: the reads are dealt to the ranks, so this rank's sample is that much thinner a slice of the WHOLE input's depth (a rank of eight
    // that holds 4-fold coverage of its own counts 32-fold k-mers after the exchange)
    e.fraction = (double)s_bytes / ((double)packed_bytes * (double)std::max(nranks, 1)); e.sample_kmers = ns; e.n1 = n1; e.n2 = n2; e.n3 = n3; e.distinct_sample = ds;
    double G = 0, lam = 0, Es = (double)n1;
    if (n2 >= 64 && n3 >= 16 && (double)n2 * 2000.0 > (double)n1) {       // (a genomic component exists: more than one doubleton per 2000 singletons)
        lam = std::min(30.0, std::max(1e-3, 3.0 * (double)n3 / (double)n2));
        G = 2.0 * (double)n2 * std::exp(lam) / (lam * lam);
        G = std::min(G, (double)ds / std::max(1e-9, 1.0 - std::exp(-lam)));      // (never more genomic k-mers than the sample's distinct ones can stand for)
        // E_s is the difference of two large numbers: the genomic singletons are known to 1 / sqrt(n3) of themselves (lambda_s comes from n3 / n2, and
        // G lambda exp(-lambda) goes like 1 / lambda).  Two standard deviations are taken off: a clean, deep input must not be sent down the instance
        // path by the noise of its own sketch (-20 %), while a few per cent of error k-mers overlooked cost a few per cent at most (the plans cross there).
        const double gs = G * lam * std::exp(-lam);
        Es = std::max(0.0, (double)n1 - gs - 2.0 * gs / std::sqrt((double)n3));
    }
    e.lambda_sample = lam;
    const double D = e.fraction * G * (1.0 - std::exp(-lam / e.fraction)) + Es;
    e.distinct_per_kmer = std::min(1.0, std::max(D / (double)ns, 1.0 / 65536.0));
    e.valid = true;
    e.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (timing_enabled()) fprintf(stderr, "[hsk] plan estimate: sample %.1f MB (1/%.0f of the input), %llu chosen k-mer instances, n1 %llu n2 %llu n3 %llu distinct %llu: lambda_s %.3f, %.4f distinct per k-mer (one in %.1f), %.2f ms\n",
                                  s_bytes / 1048576.0, 1.0 / e.fraction, (unsigned long long)ns, (unsigned long long)n1, (unsigned long long)n2, (unsigned long long)n3, (unsigned long long)ds, lam, e.distinct_per_kmer, 1.0 / e.distinct_per_kmer, e.ms);
    return HSK_OK;
}

// which homopolymer k-mers this call's sketch has found more than U (and at least 2^16) copies of: bit 0 all-A, bit 1 all-C
static u32 certain_drop_mask(hsk_ctx *c)
{
    if (tune("drop_certain", 1) == 0 || c->cfg.kmer_size > 57 || !parse_fast_enabled() || c->cfg.minimizer_size > SCAN_MAX_M) return 0;
    const u64 dmin = std::max<u64>((u64)std::max(c->cfg.upper_freq, 0), 1ULL << 16);
    return (c->est.homo_at > dmin ? 1u : 0u) | (c->est.homo_cg > dmin ? 2u : 0u);
}

static int dispatch_pipeline(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                             int64_t rid_base, hsk_result *out, int attempt = 0)
{
    int rc;
    if (attempt == 0) { c->combine_left_now = false; c->pair_cap_full = false; }
    if (attempt == 0) {
        rc = estimate_plan(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, c->comm.active() ? c->comm.nranks : 1); if (rc) return rc;
        // A k-mer with more than U copies inside the sample alone cannot be in the result: the scan leaves the instances of the all-A / all-C k-mer out
        // where they are that many (poly-A, the poly-G reads of two-colour sequencers: one bucket, one bin, one task several times the others' size).
        // From 2^16 copies on (below that nothing is gained).  Several ranks: a rank that is certain is right for all of them -- run_pipeline ORs the
        // ranks' masks in the plan's all-reduce; the general parse kernels honour the mask as well, so a rank whose parse falls back stays consistent.
        c->drop_mask_now = certain_drop_mask(c);
        if (c->drop_mask_now && timing_enabled()) fprintf(stderr, "[hsk] certain drops: the sample holds %llu copies of the all-A and %llu of the all-C k-mer (U = %d): mask %u\n",
                                                           (unsigned long long)c->est.homo_at, (unsigned long long)c->est.homo_cg, c->cfg.upper_freq, c->drop_mask_now);
    }
    c->plan_attempt = attempt;                          // (from the third attempt on run_pipeline does not consider the combining extraction at all)
    const std::vector<void *> before = c->pool.snapshot();
    const hsk_stats stats_before = c->stats;
    switch (c->nw) {
    case 1: rc = run_pipeline<1>(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, out); break;
    case 2: rc = run_pipeline<2>(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, out); break;
    default: rc = run_pipeline<3>(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, out); break;
    }
    c->vt_shift = 0;
    if (rc != HSK_OK) {
        // the failed call's kernels are drained, its result is dropped, and every device block it still holds goes back to the pool
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamSynchronize(c->d2h_stream);
        hsk_result_free(c, out);
        c->pool.release_all_but(before);
        (void)hipMemsetAsync(c->d_err, 0, 4, c->stream);
    }
    if (rc == HSK_RETRY_PLAN) {
        // the combining extraction gave up (hsk_ctx::combine_off or combine_veto is set): once more, from the reads in HBM, on the instance path;
        // the statistics describe the attempt that produces the result
        drain_profile_events(c);
        c->stats = stats_before;
        if (attempt >= 3) return fail(c, HSK_ERR_INTERNAL, "the call was started again %d times without settling on a plan", attempt);
        return dispatch_pipeline(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, out, attempt + 1);
    }
    c->plan_attempt = 0; c->est.valid = false;
    c->drop_mask_now = 0; c->dropped_now = 0;            // (the hsk_stage_* entry points parse without a plan: nothing of this call's may stay behind)
    return rc;
}

extern "C" void hsk_result_free(hsk_ctx *c, hsk_result *r)
{
    if (!r) return;
    ResultPriv *rp = (ResultPriv *)r->priv;
    if (rp) {
        for (void *p : rp->host_blocks) { if (c) c->hpool.release(p); else (void)hipHostFree(p); }
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
                        uint64_t nreads, DevInput &d, bool wait = true)
{
    DALLOC(c, d.packed, u8 *, packed_bytes + 64);
    DALLOC(c, d.roff, u64 *, (nreads + 1) * 8);
    DALLOC(c, d.rlen, u32 *, (nreads + 1) * 4);
    if (packed_bytes && packed) HIPCHK(c, hipMemcpyAsync(d.packed, packed, packed_bytes, hipMemcpyHostToDevice, c->stream));   // (null: the scan reads the host buffer in place)
    if (nreads) {
        HIPCHK(c, hipMemcpyAsync(d.roff, off, nreads * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d.rlen, len, nreads * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (wait) {
        HIPCHK(c, hipMemcpyAsync(d.roff + nreads, &packed_bytes, 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));     // host buffers (and &packed_bytes) may go away after return
    } else {
        // hsk_count returns after the pipeline's final wait: the caller's buffers outlive the copies, only the stack value needs a home
        u64 *stage = (u64 *)((char *)c->pinned + c->pinned_bytes - 256);
        *stage = packed_bytes;
        HIPCHK(c, hipMemcpyAsync(d.roff + nreads, stage, 8, hipMemcpyHostToDevice, c->stream));
    }
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

// read_byte_off[r] == sum of (read_len + 3) / 4 over the reads before r?  (what DnaBuffer guarantees, reference src/dnabuffer.cpp:24-31)
// Four threads: segment sums, then every segment is checked against its base.  Runs beside the GPU's scan.
// ... and do they end inside the packed buffer?  (A back-to-back index whose summed length runs past packed_bytes is "not fine"
// here too: the caller's offsets are then checked on the device like any other index, and index_check_kernel rejects them.)
// uniform_len != 0: ... and are all reads that long?  (the device generated the lengths from a sample instead of copying 4 bytes per read)
static bool offsets_back_to_back(const uint64_t *off, const uint32_t *len, uint64_t nreads, uint64_t packed_bytes, uint32_t uniform_len)
{
    constexpr int NT = 4;
    uint64_t seg[NT + 1], sum[NT];
    for (int t = 0; t <= NT; ++t) seg[t] = nreads * (uint64_t)t / NT;
    {
        std::thread th[NT];
        for (int t = 0; t < NT; ++t) th[t] = std::thread([&, t]() { uint64_t s = 0; for (uint64_t r = seg[t]; r < seg[t + 1]; ++r) s += ((uint64_t)len[r] + 3) >> 2; sum[t] = s; });
        for (auto &x : th) x.join();
    }
    uint64_t base[NT]; { uint64_t b = 0; for (int t = 0; t < NT; ++t) { base[t] = b; b += sum[t]; } }
    bool ok[NT];
    {
        std::thread th[NT];
        for (int t = 0; t < NT; ++t) th[t] = std::thread([&, t]() {
            uint64_t b = base[t]; bool good = true;
            for (uint64_t r = seg[t]; r < seg[t + 1]; ++r) { good &= off[r] == b; good &= uniform_len == 0 || len[r] == uniform_len; b += ((uint64_t)len[r] + 3) >> 2; }
            ok[t] = good; });
        for (auto &x : th) x.join();
    }
    bool all = true; for (int t = 0; t < NT; ++t) all &= ok[t];
    return all && base[NT - 1] + sum[NT - 1] <= packed_bytes;
}

// Inputs in pinned host memory (hsk_host_alloc, hipHostMalloc, hipHostRegister) are not copied first: scan_kernel reads the
// packed reads in place over PCIe -- once, while it hashes them -- and leaves the copy the later stages need in HBM, so the
// transfer hides behind the VALU-bound scan; only the read index (12 bytes per read) travels ahead of it.  Pageable inputs
// take staged copies.  HSK_ZERO_COPY=0 always copies.
static bool zero_copy_enabled()
{
    return tune("zero_copy", 1) != 0;
}

// hsk_count() with a pinned DnaBuffer: only the read lengths travel ahead of the scan, the byte offsets are their prefix sums
// (roff_*_kernel); the caller's offsets are compared with that layout by host threads while the GPU scans.  Returns a status
// like upload_input: the caller releases whatever was allocated, on every path.
static int derive_input(hsk_ctx *c, uint64_t packed_bytes, const uint64_t *off, const uint32_t *len, uint64_t nreads, DevInput &d, u64 *&d_given, u64 *&d_tsum)
{
    const u64 ntl = (nreads + ROFF_TILE - 1) / ROFF_TILE;
    DALLOC(c, d.packed, u8 *, packed_bytes + 64);
    DALLOC(c, d.roff, u64 *, (nreads + 1) * 8);
    DALLOC(c, d.rlen, u32 *, (nreads + 1) * 4);
    DALLOC(c, d_given, u64 *, (nreads + 1) * 8);
    DALLOC(c, d_tsum, u64 *, ntl * 8 + 64);
    u64 *stage = (u64 *)((char *)c->pinned + c->pinned_bytes - 256);
    *stage = packed_bytes;
    // Fixed-length reads (sequencer output): three samples say so, the device fills in the lengths, and the host threads that
    // compare the offsets anyway verify EVERY length while the GPU scans -- 4 bytes per read less on the link (267 MB of 2.8 GB at
    // 10 Gbp of 150-bp reads).  A single read of another length sends the call down the same road as a buffer with gaps.
    const bool uniform_enabled = tune("uniform_len", 1) != 0;
    const uint32_t ulen = (uniform_enabled && len[0] != 0 && len[0] == len[nreads / 2] && len[0] == len[nreads - 1]) ? len[0] : 0;
    if (ulen) { hipLaunchKernelGGL(rlen_fill_kernel, dim3(2048), dim3(256), 0, c->stream, d.rlen, nreads, ulen); c->rlen_host = len; c->stats.h2d_bytes -= nreads * 4; }
    else HIPCHK(c, hipMemcpyAsync(d.rlen, len, nreads * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(roff_tilesum_kernel, dim3((u32)ntl), dim3(PARSE_THREADS), 0, c->stream, d.rlen, nreads, d_tsum);
    hipLaunchKernelGGL(roff_tilescan_kernel, dim3(1), dim3(PARSE_THREADS), 0, c->stream, d_tsum, ntl);
    hipLaunchKernelGGL(roff_write_kernel, dim3((u32)ntl), dim3(PARSE_THREADS), 0, c->stream, d.rlen, nreads, d_tsum, d.roff);
    HIPCHK(c, hipMemcpyAsync(d.roff + nreads, stage, 8, hipMemcpyHostToDevice, c->stream));
    // the caller's offsets stay on the host: four threads check them against the back-to-back layout while the GPU scans
    c->roff_given = d_given; c->roff_host = off;
    c->roff_check = std::async(std::launch::async, offsets_back_to_back, off, len, nreads, packed_bytes, ulen);
    return HSK_OK;
}

extern "C" int hsk_count(hsk_ctx *c, const uint8_t *packed, uint64_t packed_bytes, const uint64_t *off, const uint32_t *len,
                         uint64_t nreads, int64_t rid_base, hsk_result *out)
{
    if (!c || !out || (nreads && (!off || !len)) || (packed_bytes && !packed)) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    enter_ctx(c);
    tmark(nullptr); tmark("hsk_count enter");
    const bool device_check = nreads >= (1u << 20);          // a serial host loop over 10^8 reads costs more than the whole count
    int rc = device_check ? HSK_OK : check_host_index(c, packed_bytes, off, len, nreads); if (rc) return rc;
    const u8 *zc = nullptr;
    if (zero_copy_enabled() && packed_bytes >= (16u << 20)) {
        hipPointerAttribute_t at; memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, packed) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) zc = (const u8 *)at.devicePointer;
        else (void)hipGetLastError();
    }
    // pinned input of some size: slab ingest (DMA copies pipelined with the scan, parse_count) instead of reads over PCIe in place;
    // HSK_H2D_SLABS=0: in place as in round 2, =n: n slabs
    const int slabs_env = (int)tune("h2d_slabs", 16);
    const bool slab_ingest = zc != nullptr && slabs_env > 1 && packed_bytes >= (32u << 20);
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    const bool derive_enabled = tune("derive_offsets", 1) != 0;
    // only the read lengths travel ahead of the scan (see roff_tilesum_kernel).  The host threads' verdict on the derived index is read by the
    // fast parse (parse_count's scan branch, parse_ingest_pipelined): with the general parse kernels from the start (M > SCAN_MAX_M, HSK_PARSE_FAST=0)
    // the caller's index travels and is checked on the device like a pageable one
    const bool derive = zc != nullptr && device_check && derive_enabled && parse_fast_enabled() && c->cfg.minimizer_size <= SCAN_MAX_M;
    DevInput d;
    u64 *d_given = nullptr, *d_tsum = nullptr;
    EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 5; (void)hipEventRecord(ep.a, c->stream); }
    if (!derive) rc = upload_input(c, zc ? nullptr : packed, packed_bytes, off, len, nreads, d, false);
    else rc = derive_input(c, packed_bytes, off, len, nreads, d, d_given, d_tsum);
    if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
    c->stats.h2d_bytes += packed_bytes + nreads * (derive ? 4 : 12);      // (derive: the offsets stay on the host; fixed-length reads: the lengths too, see derive_input)
    if (rc == HSK_OK && device_check && !derive) {
        hipLaunchKernelGGL(index_check_kernel, dim3(1024), dim3(256), 0, c->stream, d.roff, d.rlen, nreads, packed_bytes, c->d_err);
        c->index_unchecked = true;
    }
    c->zc_src = slab_ingest ? nullptr : zc;
    c->h2d_src = slab_ingest ? packed : nullptr; c->h2d_slabs = slabs_env;
    tmark(zc ? (derive ? "input enqueued (zero-copy packed, offsets derived from the lengths)" : "input enqueued (zero-copy packed)") : "input enqueued (copies)");
    if (rc == HSK_OK) rc = dispatch_pipeline(c, d.packed, packed_bytes, d.roff, d.rlen, nreads, rid_base, out);
    tmark("pipeline returned");
    c->zc_src = nullptr; c->h2d_src = nullptr; c->index_unchecked = false; c->roff_given = nullptr; c->roff_host = nullptr; c->roff_bad = false; c->rlen_host = nullptr;
    if (c->roff_check.valid()) (void)c->roff_check.get();          // (the pipeline failed before it collected the verdict)
    c->pool.release(d_given); c->pool.release(d_tsum);
    free_input(c, d);
    return rc;
}

extern "C" int hsk_count_device(hsk_ctx *c, const void *d_packed, uint64_t packed_bytes, const void *d_off, const void *d_len,
                                uint64_t nreads, int64_t rid_base, hsk_result *out)
{
    if (!c || !out || (nreads && (!d_off || !d_len)) || (packed_bytes && !d_packed)) return HSK_ERR_INVALID_ARG;
    if (((uintptr_t)d_packed & 3) != 0) return fail(c, HSK_ERR_INVALID_ARG, "d_packed must be 4-byte aligned");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    enter_ctx(c);
    // the kernels index roff[r+1]: build the (nreads+1)-entry offset array
    u64 *roff; DALLOC(c, roff, u64 *, (nreads + 1) * 8);
    if (nreads) HIPCHK(c, hipMemcpyAsync(roff, d_off, nreads * 8, hipMemcpyDeviceToDevice, c->stream));
    u64 *stage = (u64 *)((char *)c->pinned + c->pinned_bytes - 256);
    *stage = packed_bytes;
    HIPCHK(c, hipMemcpyAsync(roff + nreads, stage, 8, hipMemcpyHostToDevice, c->stream));
    int rc = dispatch_pipeline(c, (const u8 *)d_packed, packed_bytes, roff, (const u32 *)d_len, nreads, rid_base, out);
    if (timing_enabled()) fprintf(stderr, "[hsk] device pool: %.2f GB live, %.2f GB cached (mapped %.2f GB), peak live %.2f GB\n", c->pool.bytes_live / 1e9, c->pool.bytes_cached / 1e9,
                                  c->pool.bytes_mapped() / 1e9, c->pool.peak / 1e9);
    c->pool.release(roff);
    return rc;
}

extern "C" int hsk_count_loopback(hsk_ctx *c, int nranks, const uint8_t *const *packed, const uint64_t *packed_bytes, const uint64_t *const *off,
                                  const uint32_t *const *len, const uint64_t *nreads, hsk_result *outs, int32_t *owner_out, int32_t owner_capacity)
{
    if (!c || nranks < 1 || nranks > 64 || !packed || !packed_bytes || !off || !len || !nreads || !outs) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    enter_ctx(c);
    std::vector<DevInput> in(nranks);
    int rc = HSK_OK;
    for (int r = 0; r < nranks && rc == HSK_OK; ++r) {
        rc = check_host_index(c, packed_bytes[r], off[r], len[r], nreads[r]);
        if (rc == HSK_OK) rc = upload_input(c, packed[r], packed_bytes[r], off[r], len[r], nreads[r], in[r]);
    }
    u32 ntasks = 0;
    std::vector<int32_t> owner(HSK_MAX_TASKS, 0);
    const std::vector<void *> before = c->pool.snapshot();
    if (rc == HSK_OK) {
        switch (c->nw) {
        case 1: rc = run_loopback<1>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        case 2: rc = run_loopback<2>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        default: rc = run_loopback<3>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        }
    }
    if (rc == HSK_OK && owner_out) { if ((u32)owner_capacity < ntasks) rc = HSK_ERR_INVALID_ARG; else memcpy(owner_out, owner.data(), sizeof(int32_t) * ntasks); }
    for (auto &d : in) free_input(c, d);
    if (rc != HSK_OK) {
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamSynchronize(c->d2h_stream);
        for (int r = 0; r < nranks; ++r) hsk_result_free(c, &outs[r]);
        c->pool.release_all_but(before);
        (void)hipMemsetAsync(c->d_err, 0, 4, c->stream);
    }
    return rc;
}

// the same with every virtual rank's reads already resident in HBM (full-size runs of the multi-rank data path: tests/test_gpu_multirank.py,
// bench.py variants[multi_rank_path]); d_off[r] has nreads[r] entries
extern "C" int hsk_count_loopback_device(hsk_ctx *c, int nranks, const void *const *d_packed, const uint64_t *packed_bytes, const void *const *d_off,
                                         const void *const *d_len, const uint64_t *nreads, hsk_result *outs, int32_t *owner_out, int32_t owner_capacity)
{
    if (!c || nranks < 1 || nranks > 64 || !d_packed || !packed_bytes || !d_off || !d_len || !nreads || !outs) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    enter_ctx(c);
    std::vector<DevInput> in(nranks);
    std::vector<u64 *> roffs(nranks, nullptr);
    u64 *stage = (u64 *)((char *)c->pinned + (256u << 10));                   // (pinned staging: one end offset per rank)
    int rc = HSK_OK;
    for (int r = 0; r < nranks && rc == HSK_OK; ++r) {
        if (((uintptr_t)d_packed[r] & 3) != 0) { rc = fail(c, HSK_ERR_INVALID_ARG, "d_packed must be 4-byte aligned"); break; }
        roffs[r] = (u64 *)c->pool.alloc((nreads[r] + 1) * 8);
        if (!roffs[r]) { rc = fail(c, HSK_ERR_OOM, "read offsets of rank %d", r); break; }
        if (nreads[r]) HIPCHK(c, hipMemcpyAsync(roffs[r], d_off[r], nreads[r] * 8, hipMemcpyDeviceToDevice, c->stream));
        stage[r] = packed_bytes[r];
        HIPCHK(c, hipMemcpyAsync(roffs[r] + nreads[r], stage + r, 8, hipMemcpyHostToDevice, c->stream));
        in[r].packed = (u8 *)const_cast<void *>(d_packed[r]); in[r].roff = roffs[r]; in[r].rlen = (u32 *)const_cast<void *>(d_len[r]);
    }
    u32 ntasks = 0;
    std::vector<int32_t> owner(HSK_MAX_TASKS, 0);
    const std::vector<void *> before = c->pool.snapshot();
    if (rc == HSK_OK) {
        switch (c->nw) {
        case 1: rc = run_loopback<1>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        case 2: rc = run_loopback<2>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        default: rc = run_loopback<3>(c, nranks, in.data(), packed_bytes, nreads, outs, owner.data(), &ntasks); break;
        }
    }
    if (rc == HSK_OK && owner_out) { if ((u32)owner_capacity < ntasks) rc = HSK_ERR_INVALID_ARG; else memcpy(owner_out, owner.data(), sizeof(int32_t) * ntasks); }
    if (rc != HSK_OK) {
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamSynchronize(c->d2h_stream);
        for (int r = 0; r < nranks; ++r) hsk_result_free(c, &outs[r]);
        c->pool.release_all_but(before);
        (void)hipMemsetAsync(c->d_err, 0, 4, c->stream);
    }
    (void)hipStreamSynchronize(c->stream);
    for (u64 *p : roffs) c->pool.release(p);
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
    enter_ctx(c);
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
        rc = expand_task<NW>(c, ts, st.sm_len, source_from_store(st, d.packed, packed_bytes), st.sm_pos, st.sm_rid, dk, dv);
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
    enter_ctx(c);
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
    enter_ctx(c);
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
    enter_ctx(c);
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
        // the host-vector all-reduce the pipeline uses (pinned staging, one wait) and its status element
        cm.solo = true; cm.stage = c->comm.stage; cm.stage_bytes = c->comm.stage_bytes;
        std::vector<u64> v(h.begin(), h.begin() + 1000);
        if (cm.allreduce_with_status(v, RCCL_SUM, false, c->stream, c->pool) != 0 || v.size() != 1000 || !std::equal(v.begin(), v.end(), h.begin())) { out = fail(c, HSK_ERR_COMM, "staged all-reduce with status (sum) failed: %s", cm.last_error.c_str()); break; }
        if (cm.allreduce_with_status(v, RCCL_MAX, true, c->stream, c->pool) != 1 || v.size() != 1000 || !std::equal(v.begin(), v.end(), h.begin())) { out = fail(c, HSK_ERR_COMM, "staged all-reduce with status (max, failed rank) failed: %s", cm.last_error.c_str()); break; }
        std::vector<u64> big(h); big.resize(40000, 7);                       // larger than the staging area: unstaged path
        std::vector<u64> big0(big);
        if (cm.allreduce_with_status(big, RCCL_SUM, false, c->stream, c->pool) != 0 || big != big0) { out = fail(c, HSK_ERR_COMM, "unstaged all-reduce with status failed"); break; }
        cm.solo = false;
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
static int synth_reads_impl(hsk_ctx *c, uint64_t genome_len, uint32_t read_len, uint64_t nreads, uint64_t seed, uint64_t first_read, double error_rate,
                            void **d_packed, uint64_t *packed_bytes, void **d_off, void **d_len);
extern "C" int hsk_synth_reads(hsk_ctx *c, uint64_t genome_len, uint32_t read_len, uint64_t nreads, uint64_t seed, uint64_t first_read,
                               void **d_packed, uint64_t *packed_bytes, void **d_off, void **d_len)
{
    return synth_reads_impl(c, genome_len, read_len, nreads, seed, first_read, 0.0, d_packed, packed_bytes, d_off, d_len);
}
extern "C" int hsk_synth_reads_err(hsk_ctx *c, uint64_t genome_len, uint32_t read_len, uint64_t nreads, uint64_t seed, uint64_t first_read, double error_rate,
                                   void **d_packed, uint64_t *packed_bytes, void **d_off, void **d_len)
{
    if (!(error_rate >= 0.0 && error_rate <= 0.75)) return HSK_ERR_INVALID_ARG;      // 0.75: every base uniform over ACGT whatever the genome says
    return synth_reads_impl(c, genome_len, read_len, nreads, seed, first_read, error_rate, d_packed, packed_bytes, d_off, d_len);
}
static int synth_reads_impl(hsk_ctx *c, uint64_t genome_len, uint32_t read_len, uint64_t nreads, uint64_t seed, uint64_t first_read, double error_rate,
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
    const u32 err_thresh = (u32)std::min<double>(error_rate * 4294967296.0, 4294967295.0);
    if (bytes) hipLaunchKernelGGL(synth_reads_kernel, dim3((u32)std::min<u64>((bytes + 255) / 256, 1u << 22)), dim3(256), 0, c->stream, gw, genome_len, read_len, nreads, seed2 + first_read, pk, err_thresh);
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

extern "C" int hsk_format_entries(hsk_ctx *c, const void *entries, uint64_t n, int32_t nw, int32_t on_device, char *text, uint64_t capacity, uint64_t *nbytes)
{
    if (!c || !nbytes || (n && !entries) || nw < 1 || nw > 3 || nw != c->nw) return HSK_ERR_INVALID_ARG;
    *nbytes = 0;
    if (n == 0) return HSK_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t eb = (size_t)n * (nw + 1) * 8;
    u64 *d_e = (u64 *)entries, *d_own = nullptr;
    if (!on_device) { DALLOC(c, d_own, u64 *, eb); HIPCHK(c, hipMemcpyAsync(d_own, entries, eb, hipMemcpyHostToDevice, c->stream)); d_e = d_own; }
    const u64 ntiles = (n + FMT_THREADS - 1) / FMT_THREADS;
    u64 *d_tile, *d_total;
    DALLOC(c, d_tile, u64 *, ntiles * 8 + 64); DALLOC(c, d_total, u64 *, 256);
    hipLaunchKernelGGL(format_entries_kernel<false>, dim3((u32)ntiles), dim3(FMT_THREADS), 0, c->stream, d_e, n, nw, c->cfg.kmer_size, d_tile, (char *)nullptr);
    hipLaunchKernelGGL(count_scan_kernel, dim3(1), dim3(CNT_THREADS), 0, c->stream, d_tile, ntiles, d_total);
    u64 *tot = (u64 *)((char *)c->pinned + c->pinned_bytes - 128);
    HIPCHK(c, hipMemcpyAsync(tot, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *nbytes = tot[0];
    int rc = HSK_OK;
    if (text && capacity >= tot[0]) {
        char *d_text; DALLOC(c, d_text, char *, tot[0] + 64);
        hipLaunchKernelGGL(format_entries_kernel<true>, dim3((u32)ntiles), dim3(FMT_THREADS), 0, c->stream, d_e, n, nw, c->cfg.kmer_size, d_tile, d_text);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(text, d_text, tot[0], hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->pool.release(d_text);
    } else if (text || capacity) rc = fail(c, HSK_ERR_INVALID_ARG, "text capacity %llu < %llu", (unsigned long long)capacity, (unsigned long long)tot[0]);
    c->pool.release(d_tile); c->pool.release(d_total); c->pool.release(d_own);
    return rc;
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

// The yardstick for "fraction of what HBM delivers": a plain copy, 16 bytes per lane, every workgroup streaming its own
// contiguous slice (MI355X_MICROARCH.md measures 6.29 TB/s = 79 % of the 8 TB/s spec this way).  bytes read + bytes written
// per launch; the best of `iters` launches counts.
__global__ __launch_bounds__(256) void copy_peak_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, u64 n16)
{
    const u64 per = (n16 + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    u64 i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {                    // four loads in flight per lane
        const uint4 a = src[i], b = src[i + 256], c_ = src[i + 512], d = src[i + 768];
        dst[i] = a; dst[i + 256] = b; dst[i + 512] = c_; dst[i + 768] = d;
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}

extern "C" int hsk_copy_peak(hsk_ctx *c, uint64_t bytes, int iters, double *gbs)
{
    if (!c || !gbs || bytes < (1u << 20) || iters < 1) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    *gbs = 0;
    const u64 n16 = bytes / 16;
    uint4 *a, *b;
    DALLOC(c, a, uint4 *, n16 * 16); DALLOC(c, b, uint4 *, n16 * 16);
    HIPCHK(c, hipMemsetAsync(a, 0x5a, n16 * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(b, 0, n16 * 16, c->stream));
    hipEvent_t e0 = ev_get(c), e1 = ev_get(c);
    double best = 0;
    for (int grid : {2048, 4096, 8192, 16384}) {
        for (int it = 0; it < iters; ++it) {
            (void)hipEventRecord(e0, c->stream);
            hipLaunchKernelGGL(copy_peak_kernel, dim3(grid), dim3(256), 0, c->stream, a, b, n16);
            (void)hipEventRecord(e1, c->stream);
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) best = std::max(best, 2.0 * (double)n16 * 16 / (ms * 1e-3) / 1e9);
        }
    }
    ev_put(c, e0); ev_put(c, e1);
    HIPCHK(c, hipGetLastError());
    c->pool.release(a); c->pool.release(b);
    *gbs = best;
    return HSK_OK;
}

// Experiment (tools/parse_overlap.py; not part of include/hsk.h): can the placement of one set of reads run BESIDE the minimizer scan of
// another?  out_ms: [0] scan with 1024 workgroups, [1] scan with `blocks` workgroups, [2] placement alone, [3] scan(`blocks`) on the
// main stream and the placement on the second stream at the same time (start of both to end of both), [4] scan inside [3], [5] placement inside [3].
extern "C" int hsk_debug_parse_overlap(hsk_ctx *c, const void *d_packed, uint64_t packed_bytes, const void *d_off, const void *d_len, uint64_t nreads,
                                       uint32_t blocks, double *out_ms)
{
    if (!c || !out_ms) return HSK_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    enter_ctx(c);
    u64 *roff; DALLOC(c, roff, u64 *, (nreads + 1) * 8);
    HIPCHK(c, hipMemcpyAsync(roff, d_off, nreads * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(roff + nreads, &packed_bytes, 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const u32 ntasks = c->cfg.ntasks ? (u32)c->cfg.ntasks : 40;
    std::vector<u32> order(ntasks); for (u32 t = 0; t < ntasks; ++t) order[t] = t;
    hipEvent_t e[6]; for (auto &x : e) x = ev_get(c);
    auto ms = [&](hipEvent_t a, hipEvent_t b) { float f = 0; (void)hipEventElapsedTime(&f, a, b); return (double)f; };
    int rc;
    ParseJob j1, j2, j3; SupermerStore st1, st2;
    (void)hipEventRecord(e[0], c->stream);
    rc = parse_count(c, (const u8 *)d_packed, packed_bytes, roff, (const u32 *)d_len, nreads, 0, ntasks, j1); if (rc) return rc;
    (void)hipEventRecord(e[1], c->stream);
    rc = parse_place(c, j1, order, st1); if (rc) return rc;
    (void)hipEventRecord(e[2], c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out_ms[0] = ms(e[0], e[1]); out_ms[2] = ms(e[1], e[2]);
    c->scan_blocks = blocks;
    (void)hipEventRecord(e[0], c->stream);
    rc = parse_count(c, (const u8 *)d_packed, packed_bytes, roff, (const u32 *)d_len, nreads, 0, ntasks, j2);
    (void)hipEventRecord(e[1], c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (rc) { c->scan_blocks = 0; return rc; }
    out_ms[1] = ms(e[0], e[1]);
    // both at once: the placement of j1's records (again, into a second store) on the second stream, the scan of j3 on the main stream
    (void)hipEventRecord(e[0], c->stream);
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, e[0], 0));
    hipStream_t main_s = c->stream;
    c->stream = c->comm_stream;
    (void)hipEventRecord(e[3], c->stream);
    rc = parse_place(c, j1, order, st2);
    (void)hipEventRecord(e[4], c->stream);
    c->stream = main_s;
    if (rc) { c->scan_blocks = 0; return rc; }
    rc = parse_count(c, (const u8 *)d_packed, packed_bytes, roff, (const u32 *)d_len, nreads, 0, ntasks, j3);
    (void)hipEventRecord(e[1], c->stream);
    c->scan_blocks = 0;
    HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    if (rc) return rc;
    out_ms[4] = ms(e[0], e[1]); out_ms[5] = ms(e[3], e[4]);
    out_ms[3] = std::max(ms(e[0], e[1]), ms(e[0], e[4]));
    parse_release(c, j1); parse_release(c, j2); parse_release(c, j3); free_store(c, st1); free_store(c, st2);
    c->pool.release(roff);
    for (auto x : e) ev_put(c, x);
    return HSK_OK;
}

// diagnostic build only: phase clock sums of the onesweep kernel (zeros in the product build)
extern "C" int hsk_debug_diag(unsigned long long *out, int n, int reset)
{
    if (!out || n < 1 || n > 32) return HSK_ERR_INVALID_ARG;
    memset(out, 0, sizeof(unsigned long long) * n);
#ifdef HSK_DIAG
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hsk::g_diag), sizeof(unsigned long long) * n) != hipSuccess) return HSK_ERR_HIP;
    if (n >= 32) { if (hipMemcpyFromSymbol(out + 20, HIP_SYMBOL(hsk::g_scan_diag), sizeof(unsigned long long) * 10) != hipSuccess) return HSK_ERR_HIP; }
    if (reset & 4) {                                            // agg_finish_kernel's stamps instead
        unsigned long long z[16] = {0};
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hsk::g_agg_diag), sizeof(unsigned long long) * (n < 16 ? n : 16)) != hipSuccess) return HSK_ERR_HIP;
        if (hipMemcpyToSymbol(HIP_SYMBOL(hsk::g_agg_diag), z, sizeof z) != hipSuccess) return HSK_ERR_HIP;
        return HSK_OK;
    }
    if (reset & 2) {                                            // the fused expand + scatter kernel's stamps instead
        unsigned long long z[16] = {0};
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hsk::g_xs_diag), sizeof(unsigned long long) * (n < 16 ? n : 16)) != hipSuccess) return HSK_ERR_HIP;
        if (hipMemcpyToSymbol(HIP_SYMBOL(hsk::g_xs_diag), z, sizeof z) != hipSuccess) return HSK_ERR_HIP;
        return HSK_OK;
    }
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hsk::g_diag), z, sizeof z) != hipSuccess) return HSK_ERR_HIP; }
#else
    (void)reset;
#endif
    return HSK_OK;
}
