// dnaseq.hpp -- view of one 2-bit packed read inside a DnaBuffer.
// Memory layout = ABI (reference include/dnaseq.hpp:33, src/dnaseq.cpp:9-56): base i lives in
// byte i/4 at bit shift 6-2*(i%4), codes A/a/N/n=0 C/c=1 G/g=2 T/t=3, unused tail bits are zero.
#pragma once
#include <cstddef>
#include <cstdint>
#include <ostream>
#include <string>

namespace hysortk {

class DnaSeq {
public:
    DnaSeq() = default;
    DnaSeq(size_t nbases, uint8_t *mem) : len_(nbases), mem_(mem) {}
    // packs `nbases` ASCII characters of `s` into `mem` (which must hold bytesneeded(nbases) bytes)
    DnaSeq(const char *s, size_t nbases, uint8_t *mem) : len_(nbases), mem_(mem) { pack(s); }

    size_t size() const { return len_; }
    size_t numbytes() const { return bytesneeded(len_); }
    int remainder() const { return static_cast<int>(4 * numbytes() - len_); }
    const uint8_t *data() const { return mem_; }

    int operator[](size_t i) const { return (mem_[i >> 2] >> (6 - 2 * (i & 3))) & 3; }
    int regular_at(size_t i) const { return (*this)[i]; }
    int revcomp_at(size_t i) const { return 3 - (*this)[len_ - 1 - i]; }

    std::string ascii() const
    {
        std::string out(len_, 'A');
        for (size_t i = 0; i < len_; ++i) out[i] = "ACGT"[(*this)[i]];
        return out;
    }

    bool operator==(const DnaSeq &o) const
    {
        if (len_ != o.len_) return false;
        for (size_t i = 0; i < len_; ++i) if ((*this)[i] != o[i]) return false;
        return true;
    }
    bool operator!=(const DnaSeq &o) const { return !(*this == o); }
    bool operator<(const DnaSeq &o) const
    {
        const size_t n = len_ < o.len_ ? len_ : o.len_;
        for (size_t i = 0; i < n; ++i) { int a = (*this)[i], b = o[i]; if (a != b) return a < b; }
        return false;
    }

    static size_t bytesneeded(size_t nbases) { return (nbases + 3) / 4; }
    static uint8_t getcharcode(char c)
    {
        switch (c) {
        case 'A': case 'a': case 'N': case 'n': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;      // "non-nucleotide characters cause undefined behaviour" in the reference
        }
    }
    static char getcodechar(int code) { return "ACGTX"[code]; }

    friend std::ostream &operator<<(std::ostream &os, const DnaSeq &s) { return os << s.ascii(); }

private:
    size_t len_ = 0;
    uint8_t *mem_ = nullptr;

    void pack(const char *s)
    {
        const size_t nb = numbytes();
        for (size_t b = 0; b < nb; ++b) {
            uint8_t byte = 0;
            for (size_t j = 0; j < 4 && 4 * b + j < len_; ++j)
                byte |= static_cast<uint8_t>(getcharcode(s[4 * b + j]) << (6 - 2 * j));   // code 4 spills exactly as in the reference
            mem_[b] = byte;
        }
    }
};

} // namespace hysortk
