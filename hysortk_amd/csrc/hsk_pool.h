// hsk_pool.h -- the device memory pool of a context.
//
// Device memory mapped for the FIRST time is what a process's first call pays for (the driver clears it: 17 - 60 ms per GB, tools/exp/
// malloc_cost.hip), and hysortk::kmer_count() is called once per process.  Rounds 1 - 3 kept freed blocks whole and handed one only to a
// request of nearly its size: a 10 Gbp call with 46 GB live at its peak had 89 GB mapped.  Now a freed block is a SEGMENT of the region it was
// allocated in: a request takes the smallest free segment that holds it and leaves the rest behind as a free segment of its own (best fit, split),
// a released segment joins its free neighbours of the same region (coalesce).  Regions go back to the runtime only when they are entirely free
// (trim: on an allocation failure, or when the context goes).
//
// Reuse is stream-ordered by convention (as before): whoever releases a block has enqueued its last user, whoever allocates one enqueues its
// first user on the same stream or behind an event.
//
// The backend (hipMalloc / hipFree) is a pair of function pointers so that the segment logic runs on the CPU against malloc (tests/test_pool.py).
#pragma once
#include <algorithm>
#include <cstddef>
#include <map>
#include <vector>

struct DevPool {
    typedef int (*MallocFn)(void **, size_t);
    typedef int (*FreeFn)(void *);
    MallocFn be_malloc = nullptr; FreeFn be_free = nullptr;          // set by the owner before the first allocation (0: success)
    static constexpr size_t ALIGN = 256;
    static constexpr size_t MIN_SPLIT = (size_t)1 << 20;             // a remainder below this stays with the block it was cut from

    struct Seg { size_t size; bool free; char *region; };
    std::map<char *, Seg> segs;                                      // every segment, live or free, by address
    std::multimap<size_t, char *> free_by_size;
    std::map<char *, size_t> regions;                                // what the backend gave us
    size_t bytes_live = 0, bytes_cached = 0, peak = 0;

    void unlist(std::map<char *, Seg>::iterator it)
    {
        auto r = free_by_size.equal_range(it->second.size);
        for (auto q = r.first; q != r.second; ++q) if (q->second == it->first) { free_by_size.erase(q); break; }
    }
    void *alloc(size_t bytes)
    {
        if (bytes == 0) bytes = ALIGN;
        bytes = (bytes + ALIGN - 1) & ~(ALIGN - 1);
        auto f = free_by_size.lower_bound(bytes);
        if (f != free_by_size.end()) {
            auto it = segs.find(f->second);
            free_by_size.erase(f);
            Seg &s = it->second;
            bytes_cached -= s.size;
            if (s.size - bytes >= MIN_SPLIT) {                        // the rest stays behind as a free segment
                char *rest = it->first + bytes;
                const size_t rsz = s.size - bytes;
                s.size = bytes;
                segs[rest] = Seg{rsz, true, s.region};
                free_by_size.insert({rsz, rest});
                bytes_cached += rsz;
            }
            s.free = false;
            bytes_live += s.size; peak = std::max(peak, bytes_live);
            return it->first;
        }
        void *p = nullptr;
        if (be_malloc(&p, bytes) != 0 || !p) {
            trim();
            p = nullptr;
            if (be_malloc(&p, bytes) != 0 || !p) return nullptr;
        }
        regions[(char *)p] = bytes;
        segs[(char *)p] = Seg{bytes, false, (char *)p};
        bytes_live += bytes; peak = std::max(peak, bytes_live);
        return p;
    }
    void release(void *p)
    {
        if (!p) return;
        auto it = segs.find((char *)p);
        if (it == segs.end() || it->second.free) return;
        bytes_live -= it->second.size;
        it->second.free = true;
        // free neighbours of the same region join
        auto nx = std::next(it);
        if (nx != segs.end() && nx->second.free && nx->second.region == it->second.region && it->first + it->second.size == nx->first) {
            unlist(nx); bytes_cached -= nx->second.size;
            it->second.size += nx->second.size;
            segs.erase(nx);
        }
        if (it != segs.begin()) {
            auto pv = std::prev(it);
            if (pv->second.free && pv->second.region == it->second.region && pv->first + pv->second.size == it->first) {
                unlist(pv); bytes_cached -= pv->second.size;
                pv->second.size += it->second.size;
                segs.erase(it);
                it = pv;
            }
        }
        free_by_size.insert({it->second.size, it->first});
        bytes_cached += it->second.size;
    }
    // regions that are entirely free go back to the runtime
    void trim()
    {
        for (auto r = regions.begin(); r != regions.end();) {
            auto it = segs.find(r->first);
            if (it != segs.end() && it->second.free && it->second.size == r->second) {
                unlist(it); bytes_cached -= it->second.size;
                segs.erase(it);
                (void)be_free(r->first);
                r = regions.erase(r);
            } else ++r;
        }
    }
    void destroy()
    {
        for (auto &r : regions) (void)be_free(r.first);
        regions.clear(); segs.clear(); free_by_size.clear(); bytes_live = bytes_cached = 0;
    }
    // Error paths return early (DALLOC / HIPCHK) without releasing what the call had allocated so far: the entry points take
    // a snapshot of the live blocks and, when the call fails, hand everything allocated since back to the pool.
    std::vector<void *> snapshot() const { std::vector<void *> v; for (auto &kv : segs) if (!kv.second.free) v.push_back(kv.first); return v; }
    void release_all_but(const std::vector<void *> &keep)        // keep: sorted (map order)
    {
        std::vector<void *> drop;
        for (auto &kv : segs) if (!kv.second.free && !std::binary_search(keep.begin(), keep.end(), (void *)kv.first)) drop.push_back(kv.first);
        for (void *p : drop) release(p);
    }
    size_t bytes_mapped() const { size_t s = 0; for (auto &r : regions) s += r.second; return s; }
};
