// hsk_plan.h -- host-side planning shared by the pipeline and the hsk_plan_* C-ABI entries.
// Pure CPU code (usable without a GPU): task count rule, heavy-hitter classification,
// task -> rank dispatch, read partitioning.  Own implementations of the reference rules:
//   prepare_supermer task count     reference src/kmerops.cpp:40-43,76
//   HeavyHitterClassifier::classify reference src/kmerops.cpp:1157-1199
//   BalancedDispatcher              reference src/kmerops.cpp:1214-1327
//   RoundRobinDispatcher            reference src/kmerops.cpp:1201-1211
//   FastaIndex::getpartition        reference src/fastaindex.cpp:52-100
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace hsk {

inline int plan_tot_tasks(int omp_max_threads, int thread_per_worker, int avg_task_per_worker, int nprocs)
{
    int avg = omp_max_threads / thread_per_worker * avg_task_per_worker - 1;   // "-1": heavy-hitter optimisation
    if (avg < 1) avg = avg_task_per_worker;
    return avg * nprocs;
}

inline void plan_classify(const uint64_t *task_kmers, int ntasks, double ratio, int32_t *types)
{
    uint64_t total = 0;
    for (int i = 0; i < ntasks; ++i) total += task_kmers[i];
    const uint64_t avg = total / (uint64_t)ntasks;
    for (int i = 0; i < ntasks; ++i) types[i] = (double)task_kmers[i] > (double)avg * ratio ? 1 : 0;
}

// One attempt of the balanced placement under the cap avg*coe.  `order` = task ids sorted by
// ascending size.  The nprocs largest tasks seed the ranks (largest -> rank 0), the remaining
// tasks are dealt from the smallest upwards to ranks nprocs-1, nprocs-2, ... skipping ranks
// that would exceed the cap.
inline bool plan_try_dispatch(const std::vector<int> &order, const uint64_t *sz, int nprocs, uint64_t cap, int32_t *owner)
{
    const int n = (int)order.size();
    std::vector<uint64_t> load(nprocs, 0);
    std::vector<int32_t> own(n, -1);
    for (int i = 0; i < nprocs; ++i) { const int idx = n - 1 - i; own[idx] = i; load[i] += sz[order[idx]]; }
    int cur = nprocs - 1;
    for (int i = 0; i < n - nprocs; ++i) {
        int tries = 0;
        for (; tries < nprocs; ++tries) {
            const bool fits = load[cur] + sz[order[i]] <= cap;
            const int r = cur;
            cur = cur == 0 ? nprocs - 1 : cur - 1;
            if (fits) { own[i] = r; load[r] += sz[order[i]]; break; }
        }
        if (tries == nprocs) return false;
    }
    for (int i = 0; i < n; ++i) owner[order[i]] = own[i];
    return true;
}

// returns 0 ok, -1 "Cannot dispatch tasks. May be too unbalanced.", -2 fewer tasks than ranks
inline int plan_dispatch(const uint64_t *task_bytes, int ntasks, int nprocs, bool plain, double upper_coe, double step, int32_t *owner)
{
    if (ntasks < 1 || nprocs < 1) return -2;
    if (plain) { for (int i = 0; i < ntasks; ++i) owner[i] = i % nprocs; return 0; }
    if (ntasks < nprocs) return -2;
    std::vector<int> order(ntasks);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return task_bytes[x] < task_bytes[y]; });
    uint64_t total = 0;
    for (int i = 0; i < ntasks; ++i) total += task_bytes[i];
    const uint64_t avg = total / (uint64_t)nprocs;
    for (double coe = 1.0 - step; coe < upper_coe; coe += step) {
        const uint64_t cap = (uint64_t)((double)avg * coe);
        if (plan_try_dispatch(order, task_bytes, nprocs, cap, owner)) return 0;
    }
    return -1;
}

// contiguous split of reads by bases; every rank but the last stops before the read that would
// take it over the average; the last rank takes the rest.  Returns -1 if a rank would get no read.
inline int plan_partition_reads(const uint64_t *read_len, uint64_t nreads, int nprocs, uint64_t *counts)
{
    uint64_t tot = 0;
    for (uint64_t i = 0; i < nreads; ++i) tot += read_len[i];
    const double avg = (double)tot / nprocs;
    uint64_t rid = 0;
    for (int p = 0; p < nprocs - 1; ++p) {
        if (rid >= nreads) return -1;
        uint64_t sofar = 0, start = rid;
        do { sofar += read_len[rid]; ++rid; } while (rid < nreads && (double)(sofar + read_len[rid]) < avg);
        counts[p] = rid - start;
    }
    counts[nprocs - 1] = nreads - rid;
    return 0;
}

} // namespace hsk
