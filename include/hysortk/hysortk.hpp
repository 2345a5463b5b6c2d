// hysortk.hpp -- the four public functions of the reference library (include/hysortk.hpp:10-16),
// implemented on top of the C ABI of libhsk.so (include/hsk.h).  Header-only: a client that was
// built against the reference switches by pointing -I at this directory and linking -lhsk instead
// of obj/libhysortk.o; the -D macros (KMER_SIZE, MINIMIZER_SIZE, LOWER/UPPER_KMER_FREQ, EXTENSION)
// keep their names and meaning.
//
//   read_dna_buffer(fasta, comm)          reference src/hysortk.cpp:18-33  (+ fastaindex.cpp)
//   kmer_count(mydna, comm)               reference src/hysortk.cpp:36-96  -> hsk_count()
//   print_kmer_histogram(list, comm)      reference src/hysortk.cpp:98-136 (64-bit bins here)
//   write_output_file(list, outdir, comm) reference src/hysortk.cpp:138-164
//
// Errors: C-ABI status codes become std::runtime_error (the reference throws / aborts).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <future>
#include <functional>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include "../hsk.h"
#include "compiletime.h"
#include "dnabuffer.hpp"
#include "kmer.hpp"

namespace hysortk {

namespace detail {

struct Ranks { int rank = 0, size = 1, local = 0; };       // local: rank among the processes of this node (one GPU each)
inline Ranks ranks_of(MPI_Comm comm)
{
    Ranks r;
#ifdef HSK_WITH_MPI
    MPI_Comm_rank(comm, &r.rank); MPI_Comm_size(comm, &r.size);
    MPI_Comm node;
    if (MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, r.rank, MPI_INFO_NULL, &node) == MPI_SUCCESS) { MPI_Comm_rank(node, &r.local); MPI_Comm_free(&node); }
    else r.local = r.rank;
#else
    (void)comm;
#endif
    return r;
}

inline void check(int status, hsk_ctx *ctx, const char *what)
{
    if (status == HSK_OK) return;
    std::string msg = std::string(what) + ": " + hsk_strerror(status);
    if (ctx && hsk_last_error(ctx)[0]) msg += std::string(" (") + hsk_last_error(ctx) + ")";
    throw std::runtime_error(msg);
}

// one hsk_ctx per process, created on first use with the compile-time configuration
inline hsk_ctx *context(MPI_Comm comm)
{
    static hsk_ctx *ctx = nullptr;
    if (ctx) return ctx;
    const Ranks r = ranks_of(comm);
    hsk_config cfg;
    hsk_config_default(&cfg);
    cfg.kmer_size = KMER_SIZE; cfg.minimizer_size = MINIMIZER_SIZE;
    cfg.lower_freq = LOWER_KMER_FREQ; cfg.upper_freq = UPPER_KMER_FREQ; cfg.extension = EXTENSION;
    // the reference's other -D macros (Makefile:1-46), where a client's build line carries them
#ifdef PLAIN_DISPATCHER
    cfg.plain_dispatcher = PLAIN_DISPATCHER;
#endif
#ifdef DISPATCH_UPPER_COE
    cfg.dispatch_upper_coe = DISPATCH_UPPER_COE;
#endif
#ifdef DISPATCH_STEP
    cfg.dispatch_step = DISPATCH_STEP;
#endif
#ifdef UNBALANCED_RATIO
    cfg.unbalanced_ratio = UNBALANCED_RATIO;
#endif
#if defined(PLAIN_CLASSIFIER) && PLAIN_CLASSIFIER
    cfg.flags |= HSK_FLAG_PLAIN_CLASSIFIER;
#endif
#if defined(SORT) && SORT == 3                                   // (not a reference value: 1 = PARADIS, 2 = RADULS there) the reference's own algorithm
    cfg.flags |= HSK_FLAG_FULL_SORT;                             // on the GPU: LSD over all key bytes + merge-count, see include/hsk.h
#endif
    int ndev = hsk_device_count();                               // GPUs this process can see (0: hsk_init reports HSK_ERR_NO_DEVICE)
    if (const char *e = std::getenv("HSK_GPUS_PER_NODE")) ndev = std::max(1, atoi(e));
    if (ndev > 0 && r.size > ndev && r.local >= ndev)
        throw std::runtime_error("more ranks on this node than GPUs: one rank per GPU (RCCL does not share a device between ranks)");
    cfg.device = ndev > 0 ? r.local % ndev : 0;                  // one rank per GPU, by node-local rank
    check(hsk_init(&cfg, &ctx), nullptr, "hsk_init");
#ifdef HSK_WITH_MPI
    if (r.size > 1) {
        char id[HSK_UNIQUE_ID_BYTES];
        if (r.rank == 0) check(hsk_comm_get_unique_id(id), ctx, "hsk_comm_get_unique_id");
        MPI_Bcast(id, HSK_UNIQUE_ID_BYTES, MPI_BYTE, 0, comm);
        check(hsk_comm_init(ctx, r.size, r.rank, id), ctx, "hsk_comm_init");
    }
#endif
    return ctx;
}

struct FaiRecord { size_t len, pos, bases, width; };            // .fai columns 2-5 (width = bytes per line incl. the line break; 0: not given, bases + 1 as the reference assumes)

} // namespace detail

namespace detail {
// read-only mapping of the byte range [a, b) of a file (the FASTA records of this rank)
struct MappedFile {
    void *base = nullptr; size_t maplen = 0, skip = 0; int fd = -1;
    MappedFile(const std::string &path, size_t a, size_t b)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        const size_t page = (size_t)sysconf(_SC_PAGESIZE), a0 = a / page * page;
        skip = a - a0; maplen = b - a0;
        base = ::mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, (off_t)a0);
        if (base == MAP_FAILED) { ::close(fd); base = nullptr; throw std::runtime_error("cannot map " + path); }
    }
    const char *text() const { return static_cast<const char *>(base) + skip; }
    ~MappedFile() { if (base) ::munmap(base, maplen); if (fd >= 0) ::close(fd); }
    MappedFile(const MappedFile &) = delete; MappedFile &operator=(const MappedFile &) = delete;
};
} // namespace detail

// FASTA + .fai -> the calling rank's DnaBuffer (contiguous run of records balanced by bases).
inline std::shared_ptr<DnaBuffer> read_dna_buffer(const std::string &fasta_fname, MPI_Comm comm)
{
    const detail::Ranks r = detail::ranks_of(comm);
    const bool timing = std::getenv("HSK_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[hysortk shim] read_dna_buffer: %-28s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<detail::FaiRecord> recs;
    {
        // The index of a read set has one line per read (6.7 M lines, 160 MB for 1 Gbp of short reads): read in one piece and parsed in place by a
        // few threads, every thread the lines that START in its share of the bytes (stream iterators took 0.4 s for it, the whole device ingest 0.3)
        const std::string fn = fasta_fname + ".fai";
        size_t total = 0;
        { std::ifstream probe(fn, std::ios::binary | std::ios::ate); if (!probe) throw std::runtime_error("cannot open " + fn); total = (size_t)probe.tellg(); }
        std::unique_ptr<detail::MappedFile> map;
        if (total) map.reset(new detail::MappedFile(fn, 0, total));
        const char *base = total ? map->text() : nullptr;
        // a line belongs to the thread whose share of the bytes it STARTS in; first the lines are counted, then every thread parses into its place
        auto first_line = [base, total](size_t lo) -> const char * {
            if (!lo) return base;
            const char *nl = static_cast<const char *>(std::memchr(base + lo - 1, '\n', total - lo + 1));
            return nl ? nl + 1 : base + total;
        };
        auto parse_range = [base, total, &first_line](size_t lo, size_t hi, detail::FaiRecord *out) -> size_t {
            const char *p = first_line(lo), *end = base + total, *stop = base + hi;
            size_t n = 0;
            while (p < stop) {
                const char *eol = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p))); if (!eol) eol = end;
                const char *q = p;
                while (q < eol && *q != '\t' && *q != ' ') ++q;                  // the name
                size_t v[4] = {0, 0, 0, 0}; int nv = 0;
                while (nv < 4) {
                    while (q < eol && (*q == '\t' || *q == ' ')) ++q;
                    if (q >= eol || *q < '0' || *q > '9') break;
                    size_t x = 0; while (q < eol && *q >= '0' && *q <= '9') x = x * 10 + (size_t)(*q++ - '0');
                    v[nv++] = x;
                }
                if (nv >= 3) { if (out) out[n] = detail::FaiRecord{v[0], v[1], v[2], nv == 4 ? v[3] : v[2] + 1}; ++n; }
                p = eol < end ? eol + 1 : end;
            }
            return n;
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nthr = total < (size_t(4) << 20) ? 1 : std::min<size_t>(hw ? hw : 4, 16);
        std::vector<size_t> cnt(nthr + 1, 0);
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<std::thread> th;
            auto work = [&](size_t t) {
                const size_t n = parse_range(total * t / nthr, total * (t + 1) / nthr, pass ? recs.data() + cnt[t] : nullptr);
                if (!pass) cnt[t + 1] = n;
            };
            for (size_t t = 1; t < nthr; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            if (!pass) { for (size_t t = 0; t < nthr; ++t) cnt[t + 1] += cnt[t]; recs.resize(cnt[nthr]); }
        }
    }
    lap("index parsed");
    std::vector<uint64_t> counts(r.size, 0);
    if (r.size > 1) {
        std::vector<uint64_t> lens(recs.size());
        for (size_t i = 0; i < recs.size(); ++i) lens[i] = recs[i].len;
        detail::check(hsk_plan_partition_reads(lens.data(), lens.size(), r.size, counts.data()), nullptr, "partition");
    } else counts[0] = recs.size();
    size_t first = 0;
    for (int p = 0; p < r.rank; ++p) first += counts[p];
    const size_t mine = counts[r.rank];
    std::vector<size_t> mylens(mine);
    bool small_len = true;
    for (size_t i = 0; i < mine; ++i) { mylens[i] = recs[first + i].len; if (mylens[i] >> 32) small_len = false; }
    // Files of some size are packed on the GPU (hsk_pack_fasta: the records' text is mapped and uploaded once, one lane per packed byte
    // gathers its four bases across the line breaks, same bytes as DnaSeq's packer incl. the code-4 spill) and come back as ONE copy into
    // the (pinned) DnaBuffer -- the host loop below packs ~50 Mbp/s, the path it feeds counts 60 Gbp/s.  HSK_HOST_INGEST=1 keeps the host loop.
    if (mine > 0) {
        const detail::FaiRecord &l = recs[first + mine - 1];
        const size_t nl = l.bases ? (l.len + l.bases - 1) / l.bases : 0;
        const size_t span0 = recs[first].pos, span1 = l.len ? l.pos + (l.bases ? (nl - 1) * l.width + (l.len - (nl - 1) * l.bases) : l.len) : l.pos;
        const char *hi = std::getenv("HSK_HOST_INGEST");
        if (span1 > span0 && span1 - span0 >= (size_t(16) << 20) && small_len && !(hi && atoi(hi) != 0) && hsk_device_count() > 0) {
            hsk_ctx *ctx = detail::context(comm);
            lap("partition, context");
            detail::MappedFile mf(fasta_fname, span0, span1);
            lap("file mapped");
            std::vector<uint64_t> pos(mine); std::vector<uint32_t> rl(mine), lb(mine), lw(mine);
            {
                const unsigned hw = std::thread::hardware_concurrency();
                const size_t nthr = mine < (size_t(1) << 20) ? 1 : std::min<size_t>(hw ? hw : 4, 8);
                auto fill = [&](size_t lo, size_t hi) {
                    for (size_t i = lo; i < hi; ++i) { const detail::FaiRecord &q = recs[first + i]; pos[i] = q.pos - span0; rl[i] = (uint32_t)q.len; lb[i] = (uint32_t)q.bases; lw[i] = (uint32_t)q.width; }
                };
                std::vector<std::thread> th;
                for (size_t t = 1; t < nthr; ++t) th.emplace_back(fill, mine * t / nthr, mine * (t + 1) / nthr);
                fill(0, mine / nthr);
                for (auto &x : th) x.join();
            }
            void *dp = nullptr, *doff = nullptr, *dlen = nullptr; uint64_t pb = 0;
            lap("record tables");
            // upload + pack on a second thread while this one allocates the DnaBuffer (pinned: the pages of 250 MB per Gbp take as long to pin as the
            // text takes to cross the link); the buffer's fill step waits for the pack and takes the bytes in one copy
            std::future<int> packed = std::async(std::launch::async, [&]() {
                return hsk_pack_fasta(ctx, mf.text(), span1 - span0, pos.data(), rl.data(), lb.data(), lw.data(), mine, &dp, &pb, &doff, &dlen);
            });
            std::shared_ptr<DnaBuffer> dbuf;
            try {
                dbuf = std::make_shared<DnaBuffer>(DnaBuffer::computebufsize(mylens), mylens, [&](uint8_t *dst) {
                    detail::check(packed.get(), ctx, "hsk_pack_fasta");
                    lap("uploaded and packed (buffer allocated meanwhile)");
                    detail::check(hsk_memcpy_d2h(ctx, dst, dp, pb), ctx, "hsk_memcpy_d2h");
                });
            } catch (...) { if (packed.valid()) packed.wait(); if (dp) hsk_synth_free(ctx, dp, doff, dlen); throw; }
            hsk_synth_free(ctx, dp, doff, dlen);
            lap("DnaBuffer built, copied");
            return dbuf;
        }
    }
    auto buf = std::make_shared<DnaBuffer>(DnaBuffer::computebufsize(mylens));
    std::ifstream fa(fasta_fname, std::ios::binary);
    if (!fa) throw std::runtime_error("cannot open " + fasta_fname);
    std::string raw, seq;
    for (size_t i = 0; i < mine; ++i) {
        const detail::FaiRecord &rec = recs[first + i];
        const size_t nlines = rec.bases ? (rec.len + rec.bases - 1) / rec.bases : 0;
        raw.resize(rec.len + nlines);
        fa.clear(); fa.seekg(static_cast<std::streamoff>(rec.pos));
        fa.read(&raw[0], static_cast<std::streamsize>(raw.size()));
        raw.resize(static_cast<size_t>(fa.gcount()));
        seq.clear();
        for (char ch : raw) if (ch != '\n' && ch != '\r' && seq.size() < rec.len) seq.push_back(ch);
        buf->push_back(seq.data(), seq.size());
    }
    return buf;
}

// The hot path.  Collective over comm.
inline std::unique_ptr<KmerListS> kmer_count(const DnaBuffer &mydna, MPI_Comm comm)
{
    hsk_ctx *ctx = detail::context(comm);
    const size_t nreads = mydna.size();
    std::vector<uint64_t> off(nreads);
    std::vector<uint32_t> len(nreads);
    for (size_t i = 0; i < nreads; ++i) { off[i] = static_cast<uint64_t>(mydna.getbufoffset(i) - mydna.data()); len[i] = static_cast<uint32_t>(mydna[i].size()); }
    int64_t rid_base = 0;
#ifdef HSK_WITH_MPI
    {
        long long n = static_cast<long long>(nreads), before = 0;
        MPI_Exscan(&n, &before, 1, MPI_LONG_LONG, MPI_SUM, comm);           // reference src/kmerops.cpp:65-71
        if (detail::ranks_of(comm).rank == 0) before = 0;
        rid_base = before;
    }
#endif
    hsk_result res;
    detail::check(hsk_count(ctx, mydna.data(), mydna.getusedbytes(), off.data(), len.data(), nreads, rid_base, &res), ctx, "hsk_count");
    std::unique_ptr<KmerListS> list(new KmerListS(res.n));
#if EXTENSION == 0
    if (res.n) std::memcpy(static_cast<void *>(list->data()), res.entries, res.n * sizeof(KmerListEntryS));
#else
    const int nw = res.nw;
    for (uint64_t i = 0; i < res.n; ++i) {
        KmerListEntryS &e = (*list)[i];
        e.kmer.CopyDataFrom(res.entries + i * (nw + 1));
        e.cnt = res.entries[i * (nw + 1) + nw];
        const uint64_t a = res.payload_off[i];
        e.pos.assign(res.pos + a, res.pos + a + e.cnt);
        e.rid.assign(res.rid + a, res.rid + a + e.cnt);
    }
#endif
    hsk_result_free(ctx, &res);
    return list;
}

inline void print_kmer_histogram(const KmerListS &kmerlist, MPI_Comm comm)
{
    uint64_t maxcount = 0;
    for (const auto &e : kmerlist) maxcount = std::max<uint64_t>(maxcount, e.cnt);
#ifdef HSK_WITH_MPI
    MPI_Allreduce(MPI_IN_PLACE, &maxcount, 1, MPI_UNSIGNED_LONG_LONG, MPI_MAX, comm);
#endif
    std::vector<unsigned long long> histo(maxcount + 1, 0);
    for (const auto &e : kmerlist) histo[e.cnt]++;
#ifdef HSK_WITH_MPI
    MPI_Allreduce(MPI_IN_PLACE, histo.data(), static_cast<int>(histo.size()), MPI_UNSIGNED_LONG_LONG, MPI_SUM, comm);
#endif
    if (detail::ranks_of(comm).rank == 0) {
        std::cout << "#count\tnumkmers" << std::endl;
        for (size_t i = 1; i < histo.size(); ++i) if (histo[i] > 0) std::cout << i << "\t" << histo[i] << std::endl;
        std::cout << std::endl;
    }
#ifdef HSK_WITH_MPI
    MPI_Barrier(comm);
#endif
}

inline void write_output_file(const KmerListS &kmerlist, const std::string &output_dir, MPI_Comm comm)
{
    const std::string fname = output_dir + "/" + std::to_string(detail::ranks_of(comm).rank) + ".out";
    std::ofstream ofs(fname, std::ios::binary);
    if (!ofs) throw std::runtime_error("Error: cannot open output file " + fname);
#if EXTENSION == 0
    // {TKmer, uint64_t} records are the C ABI's entry layout: the lines are spelled on the GPU (hsk_format_entries)
    if (!kmerlist.empty()) {
        hsk_ctx *ctx = detail::context(comm);
        uint64_t need = 0;
        detail::check(hsk_format_entries(ctx, kmerlist.data(), kmerlist.size(), (KMER_SIZE + 31) / 32, 0, nullptr, 0, &need), ctx, "hsk_format_entries");
        std::string text(static_cast<size_t>(need), '\0');
        detail::check(hsk_format_entries(ctx, kmerlist.data(), kmerlist.size(), (KMER_SIZE + 31) / 32, 0, &text[0], need, &need), ctx, "hsk_format_entries");
        ofs.write(text.data(), static_cast<std::streamsize>(text.size()));
    }
#else
    for (const auto &e : kmerlist) ofs << e.kmer << "\t" << e.cnt << "\n";
#endif
}

} // namespace hysortk
