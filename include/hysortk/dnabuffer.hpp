// dnabuffer.hpp -- owning, contiguous 2-bit read buffer: the input type of hysortk::kmer_count().
// Same observable behaviour as the reference's DnaBuffer (include/dnabuffer.hpp:14,
// src/dnabuffer.cpp:7-31): every read starts on a byte boundary, reads are stored back to back,
// the (bufsize, numreads, buf, readlens) constructor ADOPTS `buf` (released with delete[]), the
// copy constructor deep-copies.
//
// Buffers of 16 MB and more that the class allocates itself live in pinned host memory (hsk_host_alloc, include/hsk.h): the
// GPU then reads the packed reads in place while it hashes them (hsk_count's zero-copy ingest) instead of waiting for a
// staged copy.  Smaller buffers, adopted buffers and machines without a GPU use plain new[] as the reference does.
#pragma once
#include <cassert>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include "../hsk.h"
#include "dnaseq.hpp"

namespace hysortk {

class DnaBuffer {
public:
    explicit DnaBuffer(size_t bufsize) : head_(0), cap_(bufsize), buf_(allocate(bufsize, pinned_)) {}

    DnaBuffer(size_t bufsize, size_t numreads, uint8_t *buf, const size_t *readlens) : head_(0), cap_(bufsize), pinned_(false), buf_(buf)
    {
        seqs_.reserve(numreads);
        for (size_t i = 0; i < numreads; ++i) {
            seqs_.emplace_back(readlens[i], buf_ + head_);
            head_ += DnaSeq::bytesneeded(readlens[i]);
        }
    }

    // a buffer of `bufsize` bytes whose packed bytes `fill(uint8_t *)` writes in one go (read_dna_buffer: one device-to-host copy of the
    // reads hsk_pack_fasta packed on the GPU); the reads' views follow from their lengths (back to back, every read on a byte boundary)
    template <class Fill>
    DnaBuffer(size_t bufsize, const std::vector<size_t> &readlens, Fill &&fill) : head_(0), cap_(bufsize), buf_(allocate(bufsize, pinned_))
    {
        fill(buf_);
        seqs_.reserve(readlens.size());
        for (size_t l : readlens) { seqs_.emplace_back(l, buf_ + head_); head_ += DnaSeq::bytesneeded(l); }
        assert(head_ <= cap_);
    }

    DnaBuffer(const DnaBuffer &o) : head_(o.head_), cap_(o.cap_), buf_(allocate(o.cap_, pinned_))
    {
        std::memcpy(buf_, o.buf_, o.cap_);
        seqs_.reserve(o.size());
        size_t off = 0;
        for (size_t i = 0; i < o.size(); ++i) { seqs_.emplace_back(o[i].size(), buf_ + off); off += o[i].numbytes(); }
    }
    DnaBuffer &operator=(const DnaBuffer &) = delete;
    ~DnaBuffer() { if (pinned_) hsk_host_free(buf_); else delete[] buf_; }
    bool pinned() const { return pinned_; }

    void push_back(const char *s, size_t len)
    {
        const size_t nb = DnaSeq::bytesneeded(len);
        assert(head_ + nb <= cap_);
        seqs_.emplace_back(s, len, buf_ + head_);
        head_ += nb;
    }

    size_t size() const { return seqs_.size(); }
    size_t getbufsize() const { return cap_; }
    size_t getusedbytes() const { return head_; }
    const uint8_t *getbufoffset(size_t i) const { return seqs_[i].data(); }
    const uint8_t *data() const { return buf_; }
    const DnaSeq &operator[](size_t i) const { return seqs_[i]; }
    size_t getrangebufsize(size_t start, size_t count) const
    {
        if (count == 0) return 0;
        const DnaSeq &last = seqs_[start + count - 1];
        return static_cast<size_t>((last.data() + last.numbytes()) - seqs_[start].data());
    }

    // every read as ASCII, one per line (reference src/dnabuffer.cpp:42-52)
    std::string getasciifilecontents() const
    {
        std::string out;
        size_t total = 0;
        for (const DnaSeq &s : seqs_) total += s.size() + 1;
        out.reserve(total);
        for (const DnaSeq &s : seqs_) { out += s.ascii(); out.push_back('\n'); }
        return out;
    }

    static size_t computebufsize(const std::vector<size_t> &seqlens)
    {
        size_t n = 0;
        for (size_t l : seqlens) n += DnaSeq::bytesneeded(l);
        return n;
    }

private:
    static uint8_t *allocate(size_t n, bool &pinned)
    {
        pinned = false;
        if (n >= (size_t(16) << 20)) { void *p = hsk_host_alloc(n); if (p) { pinned = true; return static_cast<uint8_t *>(p); } }
        return new uint8_t[n ? n : 1];
    }
    size_t head_;
    const size_t cap_;
    bool pinned_ = false;            // (declared before buf_: allocate() sets it)
    uint8_t *buf_;
    std::vector<DnaSeq> seqs_;
};

} // namespace hysortk
