// fakerccl.cpp -- TEST-ONLY stand-in for the nine RCCL entry points hsk_comm.h binds with dlsym (ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString),
// so that the N > 1 product path (process_rank with a live Comm, post_exchange against a concurrent peer,
// allreduce_with_status, the heavy-hitter list exchange, failing together) runs as N fresh processes on ONE GPU:
// RCCL itself refuses two ranks on one device.  Selected with HSK_RCCL_LIB=<this library> (hsk_comm.h honours the
// variable only when it is set); never part of the product, never loaded by default.
//
// Transport: a POSIX shared-memory segment named by the unique id; one bounded ring of NSLOT x CHUNK bytes per directed
// pair of ranks.  Messages between a pair match in issue order, byte counts must agree (like NCCL); a slot carries
// {bytes of the whole message, bytes in this slot}.  Sender: hipMemcpy device -> slot, publish; receiver: hipMemcpy
// slot -> device, release.
//
// Ordering: every operation (a lone call or a whole ncclGroupStart..End group) first drains the stream it was issued
// on, then moves its data with blocking copies on the calling thread, and returns when its own sends are consumed and
// its receives have landed.  Work enqueued on the stream afterwards is therefore ordered after the transfer, as with
// RCCL; the host simply blocks longer than it would there.  (The asynchronous stream chaining of the exchange -- events
// between the communication and the main stream -- is exercised by hsk_count_loopback with device copies.)
//
// A rank that waits longer than HSK_FAKERCCL_TIMEOUT seconds (default 60) without any byte moving raises the segment's
// abort flag: every rank then gets ncclSystemError from its current and all later calls instead of hanging.
#ifdef FAKERCCL_NO_HIP
// CPU build for tests/test_fakerccl_cpu.py: "device" buffers are host buffers, so the ring protocol, the matching rules and the
// time-outs can be exercised by processes without a GPU
#include <cstring>
typedef void *hipStream_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
static inline int hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline int hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#else
#include <hip/hip_runtime_api.h>
#endif
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

constexpr int MAXR = 8;
constexpr int NSLOT = 2;
constexpr size_t CHUNK = 4u << 20;
constexpr uint32_t MAGIC = 0x68736b66;      // "hskf"

enum { FR_OK = 0, FR_HIP = 1, FR_SYSTEM = 2, FR_INTERNAL = 3, FR_INVALID_ARG = 4, FR_INVALID_USAGE = 5 };

struct SlotHdr { uint64_t msg_bytes, nbytes; };
struct Ring {                                  // src -> dst
    std::atomic<uint64_t> head;                // slots published by the sender
    std::atomic<uint64_t> tail;                // slots released by the receiver
    SlotHdr hdr[NSLOT];
    char pad[64];
};
struct Shm {
    std::atomic<uint32_t> magic, nranks, arrived, left, abort_flag;
    char pad[64];
    Ring ring[MAXR * MAXR];
    // followed by nranks * nranks * NSLOT * CHUNK bytes of slot data
};

struct FakeComm {
    Shm *shm = nullptr; char *data = nullptr; size_t map_bytes = 0;
    int nranks = 0, rank = 0;
    char name[64] = {0};
    double timeout_s = 60;
    Ring &ring(int src, int dst) { return shm->ring[src * MAXR + dst]; }
    char *slot(int src, int dst, uint64_t idx) { return data + (((size_t)src * nranks + dst) * NSLOT + idx % NSLOT) * CHUNK; }
};

struct Op { int kind; int peer; char *buf; size_t bytes; bool host; size_t done = 0; bool announced = false; };

struct UniqueId { char internal[128]; };

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local FakeComm *g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;
thread_local bool g_have_stream = false;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

size_t dtype_size(int dt)
{
    switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 0; }
}

// Runs a set of operations to completion: all of them progress together (both directions of every pair), the
// operations of one directed pair in issue order.
int progress(FakeComm *cm, std::vector<Op> &ops)
{
    const int nr = cm->nranks, me = cm->rank;
    std::vector<std::vector<size_t>> sendq(nr), recvq(nr);
    for (size_t i = 0; i < ops.size(); ++i) (ops[i].kind == 0 ? sendq : recvq)[ops[i].peer].push_back(i);
    std::vector<size_t> si(nr, 0), ri(nr, 0);
    size_t open = ops.size();
    double last = now_s();
    while (open) {
        bool moved = false;
        if (cm->shm->abort_flag.load(std::memory_order_acquire)) return FR_SYSTEM;
        for (int p = 0; p < nr; ++p) {
            // next send to p
            while (si[p] < sendq[p].size()) {
                Op &o = ops[sendq[p][si[p]]];
                Ring &r = cm->ring(me, p);
                const uint64_t h = r.head.load(std::memory_order_relaxed), t = r.tail.load(std::memory_order_acquire);
                if (o.bytes == 0 && o.announced) { ++si[p]; --open; moved = true; continue; }
                if (h - t >= NSLOT) break;                                   // ring full: the peer has to consume first
                const size_t n = std::min(CHUNK, o.bytes - o.done);
                char *sl = cm->slot(me, p, h);
                if (n) {
                    if (o.host) memcpy(sl, o.buf + o.done, n);
                    else if (hipMemcpy(sl, o.buf + o.done, n, hipMemcpyDeviceToHost) != hipSuccess) { cm->shm->abort_flag.store(1); return FR_HIP; }
                }
                r.hdr[h % NSLOT].msg_bytes = o.bytes; r.hdr[h % NSLOT].nbytes = n;
                r.head.store(h + 1, std::memory_order_release);
                o.done += n; o.announced = true; moved = true;
                if (o.done == o.bytes) { ++si[p]; --open; }
            }
            // next receive from p
            while (ri[p] < recvq[p].size()) {
                Op &o = ops[recvq[p][ri[p]]];
                Ring &r = cm->ring(p, me);
                const uint64_t t = r.tail.load(std::memory_order_relaxed), h = r.head.load(std::memory_order_acquire);
                if (h == t) break;                                           // nothing published yet
                const SlotHdr hd = r.hdr[t % NSLOT];
                if (hd.msg_bytes != o.bytes || hd.nbytes > o.bytes - o.done) {
                    fprintf(stderr, "fakerccl[%d]: message from rank %d has %llu bytes, the matching receive %llu\n", me, p,
                            (unsigned long long)hd.msg_bytes, (unsigned long long)o.bytes);
                    cm->shm->abort_flag.store(1); return FR_INVALID_USAGE;
                }
                char *sl = cm->slot(p, me, t);
                if (hd.nbytes) {
                    if (o.host) memcpy(o.buf + o.done, sl, hd.nbytes);
                    else if (hipMemcpy(o.buf + o.done, sl, hd.nbytes, hipMemcpyHostToDevice) != hipSuccess) { cm->shm->abort_flag.store(1); return FR_HIP; }
                }
                r.tail.store(t + 1, std::memory_order_release);
                o.done += hd.nbytes; moved = true;
                if (o.done == o.bytes) { ++ri[p]; --open; }
            }
        }
        if (moved) { last = now_s(); continue; }
        if (now_s() - last > cm->timeout_s) {
            fprintf(stderr, "fakerccl[%d]: no progress for %.0f s (%zu operations open): a peer is not taking part -- aborting the communicator\n",
                    me, cm->timeout_s, open);
            cm->shm->abort_flag.store(1);
            return FR_SYSTEM;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    // a send is complete when the peer has taken it out of the ring (so that the caller may reuse the buffer AND a
    // later message cannot overtake); wait for the rings this call wrote to drain
    for (int p = 0; p < nr; ++p) {
        if (sendq[p].empty()) continue;
        Ring &r = cm->ring(me, p);
        while (r.tail.load(std::memory_order_acquire) != r.head.load(std::memory_order_relaxed)) {
            if (cm->shm->abort_flag.load(std::memory_order_acquire)) return FR_SYSTEM;
            if (now_s() - last > cm->timeout_s) {
                fprintf(stderr, "fakerccl[%d]: rank %d does not take its messages (%.0f s) -- aborting the communicator\n", me, p, cm->timeout_s);
                cm->shm->abort_flag.store(1);
                return FR_SYSTEM;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    return FR_OK;
}

int run_ops(FakeComm *cm, std::vector<Op> &ops, hipStream_t s)
{
    if (!cm || !cm->shm) return FR_INVALID_ARG;
    if (cm->shm->abort_flag.load()) return FR_SYSTEM;
    if (hipStreamSynchronize(s) != hipSuccess) return FR_HIP;             // everything issued on the stream before this call has run
    for (auto &o : ops) if (o.peer < 0 || o.peer >= cm->nranks) return FR_INVALID_ARG;
    return progress(cm, ops);
}

} // namespace

extern "C" {

int ncclGetUniqueId(UniqueId *id)
{
    if (!id) return FR_INVALID_ARG;
    memset(id->internal, 0, sizeof id->internal);
    std::random_device rd;
    snprintf(id->internal, sizeof id->internal, "/hsk_fakerccl_%d_%08x%08x", (int)getpid(), (unsigned)rd(), (unsigned)rd());
    return FR_OK;
}

int ncclCommInitRank(void **out, int nranks, UniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks || id.internal[0] != '/') return FR_INVALID_ARG;
    FakeComm *cm = new FakeComm();
    cm->nranks = nranks; cm->rank = rank;
    memcpy(cm->name, id.internal, sizeof cm->name - 1);
    if (const char *t = getenv("HSK_FAKERCCL_TIMEOUT")) cm->timeout_s = atof(t) > 0 ? atof(t) : 60;
    const size_t bytes = sizeof(Shm) + (size_t)nranks * nranks * NSLOT * CHUNK;
    const int fd = shm_open(cm->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete cm; return FR_SYSTEM; }
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); delete cm; return FR_SYSTEM; }   // (same size from every rank; new pages read as zero)
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete cm; return FR_SYSTEM; }
    cm->shm = (Shm *)m; cm->data = (char *)m + sizeof(Shm); cm->map_bytes = bytes;
    uint32_t expect = 0;
    if (cm->shm->magic.compare_exchange_strong(expect, MAGIC)) cm->shm->nranks.store((uint32_t)nranks);
    cm->shm->arrived.fetch_add(1);
    const double t0 = now_s();
    while (cm->shm->arrived.load() < (uint32_t)nranks) {                   // ncclCommInitRank is collective
        if (now_s() - t0 > cm->timeout_s) { fprintf(stderr, "fakerccl[%d]: only %u of %d ranks arrived\n", rank, cm->shm->arrived.load(), nranks); munmap(m, bytes); shm_unlink(cm->name); delete cm; return FR_SYSTEM; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (cm->shm->nranks.load() != (uint32_t)nranks) { munmap(m, bytes); delete cm; return FR_INVALID_ARG; }
    *out = cm;
    return FR_OK;
}

int ncclCommDestroy(void *c)
{
    FakeComm *cm = (FakeComm *)c;
    if (!cm) return FR_INVALID_ARG;
    if (cm->shm) {
        const uint32_t gone = cm->shm->left.fetch_add(1) + 1;
        const bool last = gone >= (uint32_t)cm->nranks;
        munmap(cm->shm, cm->map_bytes);
        if (last) shm_unlink(cm->name);
    }
    delete cm;
    return FR_OK;
}

int ncclGroupStart() { ++g_depth; return FR_OK; }

int ncclGroupEnd()
{
    if (g_depth <= 0) return FR_INVALID_USAGE;
    if (--g_depth > 0) return FR_OK;
    int rc = FR_OK;
    if (!g_ops.empty()) rc = run_ops(g_comm, g_ops, g_stream);
    g_ops.clear(); g_comm = nullptr; g_have_stream = false;
    return rc;
}

static int p2p(int kind, void *buf, size_t count, int dt, int peer, void *c, hipStream_t s)
{
    const size_t es = dtype_size(dt);
    if (!c || !es || (count && !buf)) return FR_INVALID_ARG;
    Op o{kind, peer, (char *)buf, count * es, false};
    if (g_depth > 0) {
        if (g_comm && g_comm != c) return FR_INVALID_USAGE;                  // one communicator per group is all the product uses
        if (g_have_stream && g_stream != s) return FR_INVALID_USAGE;         // ... and one stream
        g_comm = (FakeComm *)c; g_stream = s; g_have_stream = true;
        g_ops.push_back(o);
        return FR_OK;
    }
    std::vector<Op> one{o};
    return run_ops((FakeComm *)c, one, s);
}

int ncclSend(const void *buf, size_t count, int dt, int peer, void *c, hipStream_t s) { return p2p(0, (void *)buf, count, dt, peer, c, s); }
int ncclRecv(void *buf, size_t count, int dt, int peer, void *c, hipStream_t s) { return p2p(1, buf, count, dt, peer, c, s); }

int ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, int dt, int op, void *c, hipStream_t s)
{
    FakeComm *cm = (FakeComm *)c;
    if (!cm || (count && (!sendbuf || !recvbuf))) return FR_INVALID_ARG;
    if (dt != 4 && dt != 5) return FR_INVALID_ARG;                            // 64-bit integers are all the product reduces
    if (op != 0 && op != 2 && op != 3) return FR_INVALID_ARG;                 // sum, max, min
    if (g_depth > 0) return FR_INVALID_USAGE;
    if (cm->shm->abort_flag.load()) return FR_SYSTEM;
    if (hipStreamSynchronize(s) != hipSuccess) return FR_HIP;
    const int nr = cm->nranks, me = cm->rank;
    std::vector<uint64_t> mine(count);
    if (count && hipMemcpy(mine.data(), sendbuf, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return FR_HIP;
    std::vector<std::vector<uint64_t>> theirs(nr);
    std::vector<Op> ops;
    for (int p = 0; p < nr; ++p) {
        if (p == me) continue;
        theirs[p].resize(count);
        ops.push_back(Op{0, p, (char *)mine.data(), count * 8, true});
        ops.push_back(Op{1, p, (char *)theirs[p].data(), count * 8, true});
    }
    int rc = ops.empty() ? FR_OK : progress(cm, ops);
    if (rc) return rc;
    std::vector<uint64_t> acc = mine;
    for (int p = 0; p < nr; ++p) {
        if (p == me) continue;
        for (size_t i = 0; i < count; ++i) {
            const uint64_t a = acc[i], b = theirs[p][i];
            if (op == 0) acc[i] = a + b;
            else if (dt == 5) acc[i] = op == 2 ? (a > b ? a : b) : (a < b ? a : b);
            else acc[i] = op == 2 ? (uint64_t)((int64_t)a > (int64_t)b ? (int64_t)a : (int64_t)b) : (uint64_t)((int64_t)a < (int64_t)b ? (int64_t)a : (int64_t)b);
        }
    }
    if (count && hipMemcpy(recvbuf, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return FR_HIP;
    return FR_OK;
}

const char *ncclGetErrorString(int rc)
{
    switch (rc) {
    case FR_OK: return "no error (fakerccl)";
    case FR_HIP: return "unhandled HIP error (fakerccl)";
    case FR_SYSTEM: return "system error: a peer stopped taking part or the communicator was aborted (fakerccl)";
    case FR_INTERNAL: return "internal error (fakerccl)";
    case FR_INVALID_ARG: return "invalid argument (fakerccl)";
    case FR_INVALID_USAGE: return "invalid usage: message sizes or call order do not match between ranks (fakerccl)";
    default: return "unknown result code (fakerccl)";
    }
}

} // extern "C"
