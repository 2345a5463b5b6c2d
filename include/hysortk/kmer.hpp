// kmer.hpp -- k-mer key type and result records of the drop-in surface.
// Layouts are the ABI (reference include/kmer.hpp:22-83, :343-410): Kmer<N> is N uint64 words,
// base i in word i/32 at shift 2*(31 - i%32) (left-aligned, unused low bits zero);
// KmerListEntryS = { TKmer kmer; uint64_t cnt; [std::vector<PosInRead> pos; std::vector<ReadId> rid;] }.
// libhsk.so returns entries in exactly the EXTENSION==0 layout, so the shim memcpy's them.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <ostream>
#include <string>
#include <vector>
#include "compiletime.h"
#include "dnaseq.hpp"

namespace hysortk {

namespace detail {
// MurmurHash3 x64_128, seed 313, low word -- what the reference's GetHash returns
// (src/hashfuncs.cpp:42-114,233).  Generic over the key length (8, 16, 24 bytes here).
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }
inline uint64_t murmur3_64(const uint64_t *w, int nwords)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = 313, h2 = 313;
    int i = 0;
    for (; i + 2 <= nwords; i += 2) {
        uint64_t k1 = w[i] * c1; k1 = rotl(k1, 31) * c2; h1 ^= k1; h1 = rotl(h1, 27) + h2; h1 = h1 * 5 + 0x52dce729;
        uint64_t k2 = w[i + 1] * c2; k2 = rotl(k2, 33) * c1; h2 ^= k2; h2 = rotl(h2, 31) + h1; h2 = h2 * 5 + 0x38495ab5;
    }
    if (i < nwords) { uint64_t k1 = w[i] * c1; k1 = rotl(k1, 31) * c2; h1 ^= k1; }
    const uint64_t len = 8ULL * nwords;
    h1 ^= len; h2 ^= len; h1 += h2; h2 += h1; h1 = fmix(h1); h2 = fmix(h2);
    return h1 + h2;
}
// reverse the 32 two-bit groups of a word and complement them
inline uint64_t revcomp_word(uint64_t x)
{
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((x & 0x0f0f0f0f0f0f0f0fULL) << 4);
    return ~__builtin_bswap64(x);
}
} // namespace detail

template <int NLONGS>
class Kmer {
public:
    static_assert(NLONGS >= 1 && NLONGS <= 3, "K up to 95");
    static constexpr int NBYTES = 8 * NLONGS;
    typedef std::array<uint64_t, NLONGS> MERARR;

    Kmer() : longs{} {}
    explicit Kmer(const DnaSeq &s) : longs{} { for (int i = 0; i < KMER_SIZE; ++i) put(i, static_cast<uint64_t>(s[i])); }
    explicit Kmer(const char *s) : longs{} { for (int i = 0; i < KMER_SIZE; ++i) put(i, DnaSeq::getcharcode(s[i])); }
    explicit Kmer(const void *mem) { std::memcpy(longs.data(), mem, NBYTES); }

    std::string GetString() const
    {
        std::string s(KMER_SIZE, 'A');
        for (int i = 0; i < KMER_SIZE; ++i) s[i] = "ACGT"[(longs[i / 32] >> (2 * (31 - i % 32))) & 3];
        return s;
    }
    bool operator<(const Kmer &o) const { return longs < o.longs; }      // word 0 first = ACGT-lexicographic
    bool operator==(const Kmer &o) const { return longs == o.longs; }
    bool operator!=(const Kmer &o) const { return !(*this == o); }

    Kmer GetExtension(int code) const
    {
        Kmer e;
        for (int i = 0; i < NLONGS; ++i) e.longs[i] = (longs[i] << 2) | (i + 1 < NLONGS ? longs[i + 1] >> 62 : 0);
        e.longs[NLONGS - 1] |= static_cast<uint64_t>(code) << (64 * NLONGS - 2 * KMER_SIZE);
        return e;
    }
    Kmer GetTwin() const
    {
        Kmer t;
        for (int l = 0; l < NLONGS; ++l) t.longs[NLONGS - 1 - l] = detail::revcomp_word(longs[l]);
        const int sh = 64 * NLONGS - 2 * KMER_SIZE;      // in (0, 64)
        for (int i = 0; i < NLONGS; ++i) t.longs[i] = (t.longs[i] << sh) | (i + 1 < NLONGS ? t.longs[i + 1] >> (64 - sh) : 0);
        return t;
    }
    Kmer GetRep() const { Kmer t = GetTwin(); return t < *this ? t : *this; }
    uint64_t GetHash() const { return detail::murmur3_64(longs.data(), NLONGS); }

    const void *GetBytes() const { return longs.data(); }
    int getByte(int &i) const { return reinterpret_cast<const uint8_t *>(longs.data())[i]; }
    void CopyDataInto(void *mem) const { std::memcpy(mem, longs.data(), NBYTES); }
    void CopyDataFrom(const void *mem) { std::memcpy(longs.data(), mem, NBYTES); }

    static std::vector<Kmer> GetKmers(const DnaSeq &s)
    {
        std::vector<Kmer> out;
        const long n = static_cast<long>(s.size()) - KMER_SIZE + 1;
        if (n <= 0) return out;
        out.reserve(n);
        out.emplace_back(s);
        for (long i = 1; i < n; ++i) out.push_back(out.back().GetExtension(s[i + KMER_SIZE - 1]));
        return out;
    }
    static std::vector<Kmer> GetRepKmers(const DnaSeq &s)
    {
        std::vector<Kmer> out = GetKmers(s);
        for (auto &k : out) k = k.GetRep();
        return out;
    }

    friend std::ostream &operator<<(std::ostream &os, const Kmer &k) { return os << k.GetString(); }

private:
    MERARR longs;
    void put(int i, uint64_t code) { longs[i / 32] |= code << (2 * (31 - i % 32)); }
};

using TKmer = Kmer<(KMER_SIZE + 31) / 32>;

typedef uint32_t PosInRead;
typedef int32_t ReadId;

struct KmerListEntryS {
    TKmer kmer;
    uint64_t cnt = 0;
#if EXTENSION == 1
    std::vector<PosInRead> pos;
    std::vector<ReadId> rid;
#endif
    KmerListEntryS() = default;
    KmerListEntryS(TKmer k, uint64_t c) : kmer(k), cnt(c) {}
    bool operator<(const KmerListEntryS &o) const { return kmer < o.kmer; }
    bool operator==(const KmerListEntryS &o) const { return kmer == o.kmer; }
    bool operator!=(const KmerListEntryS &o) const { return kmer != o.kmer; }
    int GetByte(int &i) const { return kmer.getByte(i); }
};
#if EXTENSION == 0
static_assert(sizeof(KmerListEntryS) == sizeof(uint64_t) * ((KMER_SIZE + 31) / 32 + 1), "entry layout must equal hsk_result.entries");
#endif

typedef std::vector<KmerListEntryS> KmerListS;

} // namespace hysortk

namespace std {
template <int N> struct hash<hysortk::Kmer<N>> { size_t operator()(const hysortk::Kmer<N> &k) const { return k.GetHash(); } };
template <int N> struct less<hysortk::Kmer<N>> { bool operator()(const hysortk::Kmer<N> &a, const hysortk::Kmer<N> &b) const { return a < b; } };      // reference include/kmer.hpp:96-102
} // namespace std
