// hsk_comm.h -- the minimizer-bucket exchange across the GPUs of one node: RCCL over xGMI.
//
// Replaces the reference's MPI exchange (one process per GPU here, one MPI rank there):
//   size matrix     MPI_Alltoallv of per-task counts      reference src/kmerops.cpp:751-811
//   payload         fixed-slot MPI_Ialltoall rounds        reference src/kmerops.cpp:814-1008
//   task sizes      MPI_Reduce + MPI_Bcast for dispatch    reference src/kmerops.cpp:1287,1325
// The reference streams 80 000-byte slots through double buffers because MPI buffers live in host
// RAM; on MI355X the whole supermer store of a rank is a few GB of a 288 GB HBM, so the payload
// moves as ONE grouped send/recv per peer and array (a direct all-to-all-v: xGMI is a full mesh,
// every pair has its own link, all seven peers transfer concurrently).  Tasks are stored grouped
// by owner rank (hsk_parse.h), so each peer's share is one contiguous range: no packing kernel.
//
// RCCL is loaded lazily with dlopen so that the single-GPU path has no RCCL dependency at all and
// a process that already carries a copy (PyTorch) shares it.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "hsk_expand.h"

namespace hsk {

struct UidByValue { char internal[128]; };

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value */ UidByValue, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

constexpr int RCCL_UINT8 = 1, RCCL_UINT64 = 5, RCCL_SUM = 0, RCCL_MAX = 2;

inline RcclApi *rccl_api(std::string *err)
{
    static RcclApi api;
    static bool tried = false;
    if (api.lib) return &api;
    if (tried) { if (err) *err = "librccl not loadable"; return nullptr; }
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    // HSK_RCCL_LIB (honoured only when set): another library exporting the same nine entry points, e.g. the stand-in
    // transport of tests/fakerccl that lets the ranks of tests/test_gpu_rccl.py share ONE GPU (RCCL refuses that)
    const char *forced = getenv("HSK_RCCL_LIB");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL); if (!h) { if (err) *err = std::string("dlopen(HSK_RCCL_LIB=") + forced + ") failed: " + dlerror(); return nullptr; }
        fprintf(stderr, "[hsk] WARNING: collective transport loaded from HSK_RCCL_LIB=%s instead of librccl (a test hook: never set it in production)\n", forced);
    }
    else for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { if (err) *err = std::string("dlopen(librccl) failed: ") + dlerror(); return nullptr; }
#define HSK_SYM(field, name) *(void **)(&api.field) = dlsym(h, name); if (!api.field) { if (err) *err = std::string("missing RCCL symbol ") + name; return nullptr; }
    HSK_SYM(GetUniqueId, "ncclGetUniqueId") HSK_SYM(CommInitRank, "ncclCommInitRank") HSK_SYM(CommDestroy, "ncclCommDestroy")
    HSK_SYM(AllReduce, "ncclAllReduce") HSK_SYM(Send, "ncclSend") HSK_SYM(Recv, "ncclRecv")
    HSK_SYM(GroupStart, "ncclGroupStart") HSK_SYM(GroupEnd, "ncclGroupEnd") HSK_SYM(GetErrorString, "ncclGetErrorString")
#undef HSK_SYM
    api.lib = h;
    return &api;
}

template <typename Pool>
struct PoolBuf {
    Pool &pool; void *p;
    PoolBuf(Pool &pl, size_t bytes) : pool(pl), p(pl.alloc(bytes)) {}
    ~PoolBuf() { pool.release(p); }
};

struct Comm {
    RcclApi *api = nullptr;
    void *comm = nullptr;
    int nranks = 1, rank = 0;
    std::string last_error;

    bool solo = false;                                            // selftest: treat a one-rank communicator as active
    bool active() const { return comm != nullptr && (nranks > 1 || solo); }
    // one ncclSend / ncclRecv carries at most this many bytes; longer messages travel as several (both sides cut the same
    // byte count the same way).  80 Gbp without the per-group overlap puts ~1.9 GB per peer into one message: counts beyond
    // 2^31 are where collective libraries have broken before.  HSK_RCCL_MSG_MAX: test hook (small pieces).
    size_t msg_max = (size_t)1 << 30;

    static int get_unique_id(void *id128)
    {
        std::string e; RcclApi *a = rccl_api(&e);
        if (!a) return -1;
        return a->GetUniqueId(id128);
    }
    int init(int nranks_, int rank_, const void *id128, bool even_alone = false)
    {
        destroy();
        nranks = nranks_; rank = rank_;
        if (nranks_ == 1 && !even_alone) return 0;
        api = rccl_api(&last_error);
        if (!api) return -1;
        UidByValue id; memcpy(id.internal, id128, 128);
        if (const char *mm = getenv("HSK_RCCL_MSG_MAX")) { const long long v = atoll(mm); if (v >= 64) msg_max = (size_t)v; }
        int rc = api->CommInitRank(&comm, nranks_, id, rank_);
        if (rc) { last_error = std::string("ncclCommInitRank: ") + api->GetErrorString(rc); comm = nullptr; return rc; }
        return 0;
    }
    void destroy()
    {
        if (comm && api) api->CommDestroy(comm);
        comm = nullptr; nranks = 1; rank = 0;
    }
    int check(int rc, const char *what)
    {
        if (rc) last_error = std::string(what) + ": " + (api ? api->GetErrorString(rc) : "?");
        return rc;
    }

    // small host-side vectors (task sizes, flags): staged through `stage` (pinned, stage_bytes) when they fit, so that the
    // round trip is copy -> all-reduce -> copy -> ONE wait
    void *stage = nullptr; size_t stage_bytes = 0;
    template <typename Pool>
    int allreduce_u64(uint64_t *host, size_t n, int op, hipStream_t s, Pool &pool)
    {
        if (!active()) return 0;
        PoolBuf<Pool> b(pool, n * 8);
        if (!b.p) { last_error = "oom"; return -1; }
        const bool staged = stage != nullptr && n * 8 <= stage_bytes;
        if (staged) memcpy(stage, host, n * 8);
        if (hipMemcpyAsync(b.p, staged ? stage : (void *)host, n * 8, hipMemcpyHostToDevice, s) != hipSuccess) return -2;
        if (!staged && hipStreamSynchronize(s) != hipSuccess) return -2;
        int rc = check(api->AllReduce(b.p, b.p, n, RCCL_UINT64, op, comm, s), "ncclAllReduce");
        if (rc) return rc;
        if (hipMemcpyAsync(staged ? stage : (void *)host, b.p, n * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return -2;
        if (hipStreamSynchronize(s) != hipSuccess) return -2;
        if (staged) memcpy(host, stage, n * 8);
        return 0;
    }
    // All-reduce of `v` with one extra element: the local status.  Afterwards EVERY rank knows whether some rank has failed
    // since the last collective, and all of them leave together (a rank that returned alone would leave its peers blocked
    // in the next collective for ever).  local_failed: this rank's own status; returns < 0: RCCL error, 1: some rank failed, 0: fine.
    template <typename Pool>
    int allreduce_with_status(std::vector<uint64_t> &v, int op, bool local_failed, hipStream_t s, Pool &pool)
    {
        if (!active()) return local_failed ? 1 : 0;
        v.push_back(local_failed ? 1 : 0);
        const int rc = allreduce_u64(v.data(), v.size(), op, s, pool);
        const bool any = v.back() != 0;
        v.pop_back();
        if (rc) return rc < 0 ? rc : -3;
        return any ? 1 : 0;
    }
    template <typename Pool> int allreduce_sum_u64(uint64_t *h, size_t n, hipStream_t s, Pool &p) { return allreduce_u64(h, n, RCCL_SUM, s, p); }
    template <typename Pool> int allreduce_max_u64(uint64_t *h, size_t n, hipStream_t s, Pool &p) { return allreduce_u64(h, n, RCCL_MAX, s, p); }
};

struct ExchangeBuffers {
    uint8_t *len = nullptr; uint8_t *bytes = nullptr; uint32_t *pos = nullptr; int32_t *rid = nullptr; uint64_t nbytes = 0;
    unsigned short *sub16 = nullptr;      // the supermers' top 16 minimizer bits (combining extraction on the owner's side), beside len
    template <typename Pool> void release(Pool &pool)
    {
        pool.release(len); pool.release(bytes); pool.release(pos); pool.release(rid); pool.release(sub16);
        len = bytes = nullptr; pos = nullptr; rid = nullptr; sub16 = nullptr;
    }
};

// Host-side plan of the all-to-all-v (pure arithmetic; also used by the CPU multi-rank tests
// through hsk_plan_exchange): given the full size matrix M[src][task] = {supermers, bytes, kmers}
// and the owner table, computes what `rank` sends to / receives from every peer and where the
// (src, task) segments land in the receive arrays laid out [src][owned task ascending].
struct ExchangePlan {
    std::vector<uint64_t> send_sup, send_bytes, send_sup_off, send_byte_off;   // per peer
    std::vector<uint64_t> recv_sup, recv_bytes, recv_sup_off, recv_byte_off;   // per peer
    uint64_t recv_tot_sup = 0, recv_tot_bytes = 0;
};

// group_of / g: restrict the plan to the tasks t with (*group_of)[t] == g (the exchange proceeds in groups of
// tasks so that the transfer of group g+1 overlaps the sort of group g); group_of == nullptr plans everything.
// A rank's tasks are stored grouped by owner in ascending id, and groups are consecutive runs of an owner's
// tasks, so a (peer, group) share is still one contiguous range of the send arrays.
inline void plan_exchange(int nranks, int rank, uint32_t ntasks, const std::vector<int32_t> &owner, const std::vector<uint32_t> &order,
                          const std::vector<uint64_t> &M /* [nranks][ntasks][3] */, const std::vector<uint64_t> &task_base /* mine [ntasks][3] */,
                          ExchangePlan &pl, std::vector<TaskSegs> &segs, const std::vector<int32_t> *group_of = nullptr, int g = -1)
{
    auto in_group = [&](uint32_t t) { return group_of == nullptr || (*group_of)[t] == g; };
    pl.send_sup.assign(nranks, 0); pl.send_bytes.assign(nranks, 0); pl.send_sup_off.assign(nranks, 0); pl.send_byte_off.assign(nranks, 0);
    pl.recv_sup.assign(nranks, 0); pl.recv_bytes.assign(nranks, 0); pl.recv_sup_off.assign(nranks, 0); pl.recv_byte_off.assign(nranks, 0);
    const uint64_t *mine = &M[(size_t)rank * ntasks * 3];
    std::vector<char> seen(nranks, 0);
    for (uint32_t i = 0; i < ntasks; ++i) {                 // storage order: grouped by owner
        const uint32_t t = order[i]; const int q = owner[t];
        if (!in_group(t)) continue;
        if (!seen[q]) { seen[q] = 1; pl.send_sup_off[q] = task_base[3 * t]; pl.send_byte_off[q] = task_base[3 * t + 1]; }
        pl.send_sup[q] += mine[3 * t]; pl.send_bytes[q] += mine[3 * t + 1];
    }
    uint64_t so = 0, bo = 0;
    for (int p = 0; p < nranks; ++p) {
        pl.recv_sup_off[p] = so; pl.recv_byte_off[p] = bo;
        for (uint32_t t = 0; t < ntasks; ++t) if (owner[t] == rank && in_group(t)) { pl.recv_sup[p] += M[((size_t)p * ntasks + t) * 3]; pl.recv_bytes[p] += M[((size_t)p * ntasks + t) * 3 + 1]; }
        so += pl.recv_sup[p]; bo += pl.recv_bytes[p];
    }
    pl.recv_tot_sup = so; pl.recv_tot_bytes = bo;
    if (group_of == nullptr) segs.assign(ntasks, TaskSegs());
    std::vector<uint64_t> cs(nranks), cb(nranks);
    for (int p = 0; p < nranks; ++p) { cs[p] = pl.recv_sup_off[p]; cb[p] = pl.recv_byte_off[p]; }
    for (uint32_t t = 0; t < ntasks; ++t) {
        if (owner[t] != rank || !in_group(t)) continue;
        segs[t] = TaskSegs();
        uint64_t koff = 0;
        for (int p = 0; p < nranks; ++p) {
            const uint64_t *m = &M[((size_t)p * ntasks + t) * 3];
            if (m[0]) { ExpSeg s; s.sup_off = cs[p]; s.n_sup = m[0]; s.byte_off = cb[p]; s.kmer_off = koff; s.tile_start = 0; segs[t].segs.push_back(s); }
            cs[p] += m[0]; cb[p] += m[1]; koff += m[2];
        }
        segs[t].nkmers = koff;
    }
}

// group index of every task inside its owner's ascending task list (groups of `group_size` tasks)
inline void assign_task_groups(int nranks, uint32_t ntasks, const std::vector<int32_t> &owner, int group_size, std::vector<int32_t> &group_of, int &ngroups)
{
    std::vector<int> seen(nranks, 0);
    group_of.assign(ntasks, 0); ngroups = 0;
    for (uint32_t t = 0; t < ntasks; ++t) { const int q = owner[t]; group_of[t] = seen[q] / group_size; ++seen[q]; ngroups = std::max(ngroups, group_of[t] + 1); }
}

// One ncclGroupStart ... ncclGroupEnd bracket of byte messages.  The group is closed on EVERY path (an early return between
// the two calls would leave the communicator inside an open group: unusable afterwards, peers blocked); after the first
// error nothing more is added, end() reports it.  Messages longer than Comm::msg_max travel in pieces.
struct P2PGroup {
    Comm &cm; int rc = 0; bool open = false; std::string first_error;
    explicit P2PGroup(Comm &c) : cm(c) { note(cm.check(cm.api->GroupStart(), "ncclGroupStart")); open = rc == 0; }
    ~P2PGroup() { if (open) (void)cm.api->GroupEnd(); }
    void note(int r) { if (r && !rc) { rc = r; first_error = cm.last_error; } }
    void send(const void *p, size_t bytes, int peer, hipStream_t s, const char *what)
    {
        for (size_t o = 0; o < bytes && !rc; o += cm.msg_max)
            note(cm.check(cm.api->Send((const char *)p + o, std::min(cm.msg_max, bytes - o), RCCL_UINT8, peer, cm.comm, s), what));
    }
    void recv(void *p, size_t bytes, int peer, hipStream_t s, const char *what)
    {
        for (size_t o = 0; o < bytes && !rc; o += cm.msg_max)
            note(cm.check(cm.api->Recv((char *)p + o, std::min(cm.msg_max, bytes - o), RCCL_UINT8, peer, cm.comm, s), what));
    }
    int end()
    {
        if (open) { open = false; note(cm.check(cm.api->GroupEnd(), "ncclGroupEnd")); }
        if (rc) cm.last_error = first_error;
        return rc;
    }
};

// the grouped send/recv of one plan on stream s (self share: device copy).  Does not synchronise.
// Every rank posts what the plan says, whether its own count has failed or not (GroupFeeder::drain_after_failure): a rank
// that stopped taking part would leave its peers blocked in their receives.
inline int post_exchange(Comm &cm, hipStream_t s, bool ext, const ExchangePlan &pl, const uint8_t *sm_len, const uint8_t *sm_bytes,
                         const uint32_t *sm_pos, const int32_t *sm_rid, ExchangeBuffers &xb, const unsigned short *sm_sub16 = nullptr)
{
    const bool sub = sm_sub16 != nullptr && xb.sub16 != nullptr;          // (every rank or none: agreed with the size matrix)
    const int nr = cm.nranks, me = cm.rank;
    {
        P2PGroup g(cm);
        for (int q = 0; q < nr && !g.rc; ++q) {
            if (q == me) continue;
            if (pl.send_sup[q]) {
                g.send(sm_len + pl.send_sup_off[q], pl.send_sup[q], q, s, "ncclSend(len)");
                g.send(sm_bytes + pl.send_byte_off[q], pl.send_bytes[q], q, s, "ncclSend(bytes)");
                if (sub) g.send(sm_sub16 + pl.send_sup_off[q], pl.send_sup[q] * 2, q, s, "ncclSend(minimizer bits)");
                if (ext) {
                    g.send(sm_pos + pl.send_sup_off[q], pl.send_sup[q] * 4, q, s, "ncclSend(pos)");
                    g.send(sm_rid + pl.send_sup_off[q], pl.send_sup[q] * 4, q, s, "ncclSend(rid)");
                }
            }
            if (pl.recv_sup[q]) {
                g.recv(xb.len + pl.recv_sup_off[q], pl.recv_sup[q], q, s, "ncclRecv(len)");
                g.recv(xb.bytes + pl.recv_byte_off[q], pl.recv_bytes[q], q, s, "ncclRecv(bytes)");
                if (sub) g.recv(xb.sub16 + pl.recv_sup_off[q], pl.recv_sup[q] * 2, q, s, "ncclRecv(minimizer bits)");
                if (ext) {
                    g.recv(xb.pos + pl.recv_sup_off[q], pl.recv_sup[q] * 4, q, s, "ncclRecv(pos)");
                    g.recv(xb.rid + pl.recv_sup_off[q], pl.recv_sup[q] * 4, q, s, "ncclRecv(rid)");
                }
            }
        }
        const int rc = g.end();
        if (rc) return rc;
    }
    if (pl.send_sup[me]) {
        if (hipMemcpyAsync(xb.len + pl.recv_sup_off[me], sm_len + pl.send_sup_off[me], pl.send_sup[me], hipMemcpyDeviceToDevice, s) != hipSuccess) return -2;
        if (hipMemcpyAsync(xb.bytes + pl.recv_byte_off[me], sm_bytes + pl.send_byte_off[me], pl.send_bytes[me], hipMemcpyDeviceToDevice, s) != hipSuccess) return -2;
        if (sub && hipMemcpyAsync(xb.sub16 + pl.recv_sup_off[me], sm_sub16 + pl.send_sup_off[me], pl.send_sup[me] * 2, hipMemcpyDeviceToDevice, s) != hipSuccess) return -2;
        if (ext) {
            if (hipMemcpyAsync(xb.pos + pl.recv_sup_off[me], sm_pos + pl.send_sup_off[me], pl.send_sup[me] * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return -2;
            if (hipMemcpyAsync(xb.rid + pl.recv_sup_off[me], sm_rid + pl.send_sup_off[me], pl.send_sup[me] * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return -2;
        }
    }
    return 0;
}

template <typename Pool>
inline int exchange_supermers(Comm &cm, hipStream_t s, Pool &pool, bool ext, int /*K*/, uint32_t ntasks, const std::vector<int32_t> &owner,
                              const std::vector<uint32_t> &order, const std::vector<uint64_t> &task_tot, const std::vector<uint64_t> &task_base,
                              const uint8_t *sm_len, const uint8_t *sm_bytes, const uint32_t *sm_pos, const int32_t *sm_rid,
                              ExchangeBuffers &xb, std::vector<TaskSegs> &segs, bool local_failed = false)
{
    const int nr = cm.nranks, me = cm.rank;
    // 1. size matrix: every rank contributes its row, the sum is the full matrix (+ the ranks' status, see allreduce_with_status)
    std::vector<uint64_t> M((size_t)nr * ntasks * 3, 0);
    if (!local_failed) for (size_t i = 0; i < (size_t)ntasks * 3; ++i) M[(size_t)me * ntasks * 3 + i] = task_tot[i];
    int rc = cm.allreduce_with_status(M, RCCL_SUM, local_failed, s, pool);
    if (rc) { if (rc > 0) cm.last_error = "a rank failed before the supermer exchange"; return rc; }
    ExchangePlan pl;
    plan_exchange(nr, me, ntasks, owner, order, M, task_base, pl, segs);
    xb.len = (uint8_t *)pool.alloc(pl.recv_tot_sup + 64);
    xb.bytes = (uint8_t *)pool.alloc(pl.recv_tot_bytes + 64); xb.nbytes = pl.recv_tot_bytes;
    if (ext) { xb.pos = (uint32_t *)pool.alloc(pl.recv_tot_sup * 4 + 64); xb.rid = (int32_t *)pool.alloc(pl.recv_tot_sup * 4 + 64); }
    {   // every rank must have its receive arrays before anybody sends
        const bool oom = !xb.len || !xb.bytes || (ext && (!xb.pos || !xb.rid));
        std::vector<uint64_t> none;
        rc = cm.allreduce_with_status(none, RCCL_MAX, oom, s, pool);
        if (oom) { cm.last_error = "oom"; return -1; }
        if (rc) { if (rc > 0) cm.last_error = "a rank ran out of memory before the supermer exchange"; return rc; }
    }
    // 2. payload: one grouped send/recv per peer and array (self: device copy)
    if ((rc = post_exchange(cm, s, ext, pl, sm_len, sm_bytes, sm_pos, sm_rid, xb))) return rc;
    if (hipStreamSynchronize(s) != hipSuccess) return -2;
    return 0;
}

} // namespace hsk
