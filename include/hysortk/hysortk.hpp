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
#include <cstdio>
#include <cstring>
#include <iterator>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
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
    std::vector<detail::FaiRecord> recs;
    {
        // (the index of a read set has one line per read: parsed in place, not line by line through string streams -- 6.7 M lines took a second that way)
        std::ifstream fai(fasta_fname + ".fai", std::ios::binary);
        if (!fai) throw std::runtime_error("cannot open " + fasta_fname + ".fai");
        std::string all((std::istreambuf_iterator<char>(fai)), std::istreambuf_iterator<char>());
        const char *p = all.data(), *end = p + all.size();
        auto number = [&](size_t &v) -> bool {
            while (p < end && (*p == '\t' || *p == ' ')) ++p;
            if (p >= end || *p < '0' || *p > '9') return false;
            size_t x = 0; while (p < end && *p >= '0' && *p <= '9') x = x * 10 + (size_t)(*p++ - '0');
            v = x; return true;
        };
        while (p < end) {
            const char *eol = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p))); if (!eol) eol = end;
            const char *line_end = eol;
            while (p < line_end && *p != '\t' && *p != ' ') ++p;                  // the name
            detail::FaiRecord rec{}; const char *save_end = end; end = line_end;
            if (number(rec.len) && number(rec.pos) && number(rec.bases)) { if (!number(rec.width)) rec.width = rec.bases + 1; recs.push_back(rec); }
            end = save_end; p = eol < end ? eol + 1 : end;
        }
    }
    std::vector<uint64_t> lens(recs.size()), counts(r.size, 0);
    for (size_t i = 0; i < recs.size(); ++i) lens[i] = recs[i].len;
    if (r.size > 1) detail::check(hsk_plan_partition_reads(lens.data(), lens.size(), r.size, counts.data()), nullptr, "partition");
    else counts[0] = recs.size();
    size_t first = 0;
    for (int p = 0; p < r.rank; ++p) first += counts[p];
    const size_t mine = counts[r.rank];
    std::vector<size_t> mylens(mine);
    for (size_t i = 0; i < mine; ++i) mylens[i] = recs[first + i].len;
    // Files of some size are packed on the GPU (hsk_pack_fasta: the records' text is mapped and uploaded once, one lane per packed byte
    // gathers its four bases across the line breaks, same bytes as DnaSeq's packer incl. the code-4 spill) and come back as ONE copy into
    // the (pinned) DnaBuffer -- the host loop below packs ~50 Mbp/s, the path it feeds counts 60 Gbp/s.  HSK_HOST_INGEST=1 keeps the host loop.
    if (mine > 0) {
        const detail::FaiRecord &l = recs[first + mine - 1];
        const size_t nl = l.bases ? (l.len + l.bases - 1) / l.bases : 0;
        const size_t span0 = recs[first].pos, span1 = l.len ? l.pos + (l.bases ? (nl - 1) * l.width + (l.len - (nl - 1) * l.bases) : l.len) : l.pos;
        const char *hi = std::getenv("HSK_HOST_INGEST");
        bool small_len = true; for (size_t i = 0; i < mine; ++i) if (mylens[i] >> 32) small_len = false;
        if (span1 > span0 && span1 - span0 >= (size_t(16) << 20) && small_len && !(hi && atoi(hi) != 0) && hsk_device_count() > 0) {
            hsk_ctx *ctx = detail::context(comm);
            detail::MappedFile mf(fasta_fname, span0, span1);
            std::vector<uint64_t> pos(mine); std::vector<uint32_t> rl(mine), lb(mine), lw(mine);
            for (size_t i = 0; i < mine; ++i) { const detail::FaiRecord &q = recs[first + i]; pos[i] = q.pos - span0; rl[i] = (uint32_t)q.len; lb[i] = (uint32_t)q.bases; lw[i] = (uint32_t)q.width; }
            void *dp = nullptr, *doff = nullptr, *dlen = nullptr; uint64_t pb = 0;
            detail::check(hsk_pack_fasta(ctx, mf.text(), span1 - span0, pos.data(), rl.data(), lb.data(), lw.data(), mine, &dp, &pb, &doff, &dlen), ctx, "hsk_pack_fasta");
            std::shared_ptr<DnaBuffer> dbuf;
            try {
                dbuf = std::make_shared<DnaBuffer>(DnaBuffer::computebufsize(mylens), mylens, [&](uint8_t *dst) { detail::check(hsk_memcpy_d2h(ctx, dst, dp, pb), ctx, "hsk_memcpy_d2h"); });
            } catch (...) { hsk_synth_free(ctx, dp, doff, dlen); throw; }
            hsk_synth_free(ctx, dp, doff, dlen);
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
