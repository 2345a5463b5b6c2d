// CPU exerciser of hysortk_amd/csrc/hsk_pool.h (the device pool's segment logic against malloc): random allocate / write / check / release
// sequences; every live block keeps its pattern, live blocks never overlap, bytes_live + bytes_cached == bytes mapped, trim returns all.
#include "../hysortk_amd/csrc/hsk_pool.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
static size_t g_mapped = 0, g_limit = (size_t)1 << 30;
static std::map<void *, size_t> g_regions;
static int be_malloc(void **p, size_t n) { if (g_mapped + n > g_limit) { *p = nullptr; return 1; } *p = std::aligned_alloc(256, n); if (!*p) return 1; g_regions[*p] = n; g_mapped += n; return 0; }
static int be_free(void *p) { g_mapped -= g_regions[p]; g_regions.erase(p); std::free(p); return 0; }
int main(int argc, char **argv)
{
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    std::mt19937_64 rng(seed);
    DevPool pool; pool.be_malloc = be_malloc; pool.be_free = be_free;
    struct Blk { unsigned char *p; size_t n; unsigned char tag; };
    std::vector<Blk> live;
    size_t fails = 0;
    for (int step = 0; step < 200000; ++step) {
        const bool do_alloc = live.empty() || (rng() % 100 < 52 && live.size() < 400);
        if (do_alloc) {
            size_t n;
            switch (rng() % 4) { case 0: n = 1 + rng() % 4096; break; case 1: n = 1 + rng() % (1u << 20); break; case 2: n = (1u << 20) + rng() % (8u << 20); break; default: n = (8u << 20) + rng() % (64u << 20); }
            unsigned char *p = (unsigned char *)pool.alloc(n);
            if (!p) { ++fails; continue; }
            if (((size_t)p & 255) != 0) { std::printf("FAIL misaligned\n"); return 1; }
            const unsigned char tag = (unsigned char)(rng() & 255);
            std::memset(p, tag, std::min<size_t>(n, 4096)); p[n - 1] = tag;
            live.push_back(Blk{p, n, tag});
        } else {
            const size_t i = rng() % live.size();
            Blk b = live[i]; live[i] = live.back(); live.pop_back();
            for (size_t q = 0; q < std::min<size_t>(b.n, 4096); ++q) if (b.p[q] != b.tag) { std::printf("FAIL pattern (step %d)\n", step); return 1; }
            if (b.p[b.n - 1] != b.tag) { std::printf("FAIL tail pattern (step %d)\n", step); return 1; }
            pool.release(b.p);
        }
        if (step % 1000 == 0) {
            if (pool.bytes_live + pool.bytes_cached != pool.bytes_mapped() || pool.bytes_mapped() != g_mapped) { std::printf("FAIL accounting %zu + %zu != %zu (%zu)\n", pool.bytes_live, pool.bytes_cached, pool.bytes_mapped(), g_mapped); return 1; }
            // segments tile their regions without gaps or overlaps
            for (auto &r : pool.regions) {
                char *at = r.first; 
                for (auto it = pool.segs.find(r.first); it != pool.segs.end() && it->second.region == r.first; ++it) { if (it->first != at) { std::printf("FAIL gap\n"); return 1; } at += it->second.size; }
                if (at != r.first + r.second) { std::printf("FAIL region not covered\n"); return 1; }
            }
            // no two adjacent free segments of one region (they must have joined)
            for (auto it = pool.segs.begin(); it != pool.segs.end(); ++it) { auto nx = std::next(it); if (nx != pool.segs.end() && it->second.free && nx->second.free && it->second.region == nx->second.region) { std::printf("FAIL uncoalesced\n"); return 1; } }
            if (pool.snapshot().size() != live.size()) { std::printf("FAIL snapshot\n"); return 1; }
        }
    }
    // release_all_but keeps exactly the named blocks
    std::vector<void *> keep; for (size_t i = 0; i < live.size(); i += 2) keep.push_back(live[i].p);
    std::sort(keep.begin(), keep.end());
    pool.release_all_but(keep);
    if (pool.snapshot() != keep) { std::printf("FAIL release_all_but\n"); return 1; }
    for (void *p : keep) pool.release(p);
    pool.trim();
    if (pool.bytes_mapped() != 0 || g_mapped != 0 || pool.bytes_cached != 0 || pool.bytes_live != 0) { std::printf("FAIL trim left %zu\n", pool.bytes_mapped()); return 1; }
    std::printf("OK seed %u, %zu allocations refused at the limit, peak live %zu\n", seed, fails, pool.peak);
    return 0;
}
