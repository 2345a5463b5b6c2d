// hsk_agg.h -- "two passes, then aggregate": the last stage of filter_kmer.
//
// Replaces the tail of sort_task + count_sorted_kmers (reference src/kmerops.cpp:1382-1445).  Kernels by key width and
// payload: agg_finish_kernel (one word; described below), agg2_finish_kernel / agg3_finish_kernel (two / three words),
// agg_ext_kernel<cap, NW> (EXTENSION: the payloads grouped by key), agg_big_kernel (8-bit bins, HSK_ONEPASS
// and the last rung of the ladder).  The probe loops of the tables are hand-written assembly (agg_count_keys,
// agg2_count_keys, agg3_count_keys): as loops of the WAVE they cost 6 scalar instructions per probe instead of the ~25 the
// compiler spends on execution masks -- the scalar unit was what agg_finish_kernel ran out of --, and for multi-word keys
// the order "claimers publish, then the others wait" must not be left to the compiler's choice of which branch runs first.
//
// The reference sorts every k-mer instance of a task completely (RADULS, 8 byte passes) and
// then scans the sorted array for runs.  Sequencing data repeats every k-mer about `coverage` times, so most of
// that sorting moves copies of the same key around.  Here only the top 16 key bits are sorted globally (two
// onesweep passes, hsk_sort.h): that cuts a 2^28-key task into 65536 prefix bins of ~4096 records, and such a
// bin holds only ~4096 / coverage DISTINCT keys.  One workgroup per bin then
//   1. streams the bin's records once from HBM and counts them in an LDS hash table (open addressing; the common
//      case "key already present" is one LDS read + one LDS atomic add),
//   2. sorts the few distinct (key, count) pairs in LDS (rank by counting up to 256 keys, bitonic beyond),
//   3. applies the [L, U] filter and writes the kept entries, in ascending key order, to the bin's own slot range
//      of a scratch buffer (no inter-workgroup dependency), with their number in bin_cnt[].
// count_scan_kernel (hsk_count.h) turns bin_cnt into offsets and agg_compact_kernel moves the entries to their
// final place, building the count histogram (print_kmer_histogram, reference src/hysortk.cpp:98-136) on the way.
// Bins are ascending in the key prefix and entries ascending inside a bin, so the list is exactly the sorted,
// filtered list the reference produces.
//
// A bin with more distinct keys than the table holds is appended to its task's overflow list; the next launch, one table
// size up, takes the listed bins (hsk_host_finish.h); a bin beyond the last table raises AG_FLAG_OVERFLOW and its task takes
// the long way (full-width passes + count_kernel).
#pragma once
#include "hsk_device.h"

namespace hsk {

constexpr int AG_THREADS = 256;
constexpr int AG_PREFIX_BITS = 16;
constexpr int AG_SHIFT = 64 - AG_PREFIX_BITS;
constexpr u32 AG_BINS = 1u << AG_PREFIX_BITS;
constexpr int AG_MAX_PROBE = 32;
constexpr u64 AG_EMPTY = ~0ULL;                   // never a canonical k-mer word (the all-T k-mer's twin, all-A, is smaller)
constexpr int AG_LDS_HIST = 256;
constexpr int AG_BATCH = 8;
constexpr int AG_LOG2CAP_SMALL = 10;             // hash table slots: 1024 (error-free reads at ~30x: ~120 distinct keys per bin) ...
constexpr int AG_LOG2CAP_MEDIUM = 11;            // ... 2048 (reads with ~1 % errors: ~900) ...
constexpr int AG_LOG2CAP_LARGE = 12;             // ... 4096 slots; a task with a bin beyond that takes the long way

enum { AG_FLAG_OVERFLOW = 1 };

struct AggTask {
    const u64 *keys; u64 n;        // sorted on the top AG_PREFIX_BITS bits
    const u64 *vals;               // weighted finish (agg_finish_kernel<cap, true>, hsk_combine.h): record i stands for vals[i] instances of its key (null: one)
    u64 *bounds;                   // [nbins + 1] first record of every prefix bin (bin_bounds_kernel)
    u64 *scratch; u32 slot_shift;  // bin b writes entry e {key, count} to scratch[((bounds[b] >> slot_shift) + e) * 2 ..]
    u32 active;
    u64 *bin_cnt;                  // [nbins] kept entries of each bin (written by whichever rung of the ladder took the bin)
    u64 *bin_off;                  // [nbins + 1] agg_scan_kernel: exclusive scan of bin_cnt, total behind the last bin (may alias bin_cnt)
    u32 *flags;                    // out: AG_FLAG_*; flags[AG_BATCH] (one word further per task): the most distinct keys any bin of the task held
    // The table ladder works BIN BY BIN: a bin with more distinct keys than this launch's table holds is appended to ovf_list
    // (a task's bins are heavy-tailed -- a bin is fed by few minimizers --, so nearly every task has a few outlier bins); the
    // next launch, one table size up, takes only the listed bins (bin_list / bin_list_n).  ovf_list == null: last rung, an
    // overflowing bin raises the task's AG_FLAG_OVERFLOW instead and the host sends the whole task the long way.
    const u32 *bin_list; const u32 *bin_list_n;
    u32 *ovf_list; u32 *ovf_n;
    const struct AggLarge *lg;     // bins of very many records counted slice by slice beforehand (agg_large_slice_kernel); null: none
};
// ---- bins of very many records ----------------------------------------------------------------------------------------------------
// A bin far beyond the usual few thousand records is nearly always ONE k-mer seen millions of times (poly-A, the poly-G reads of two-colour
// sequencers, satellites).  Its records all hit one counter of one workgroup's table, and that workgroup reads them alone: 40 M copies of the
// all-A 31-mer (5 Gbp with 2 % all-A reads) kept it busy for 108 ms while the rest of the GPU had finished the batch in 10.  Such bins are cut
// into slices of AGL_SLICE records: agg_large_list_kernel finds them (up to AGL_TABLES per task) and lists their slices, agg_large_slice_kernel
// counts every slice in an LDS table (a wave whose lanes all hold the same key sends one lane with the sum) and adds the table's few pairs to
// the bin's table in GLOBAL memory (AGL_TAB slots, atomics); the bin's own workgroup then counts that table's slots instead of the records.
// More distinct keys than a slice's or the bin's table holds (a large bin of another kind): the bin is marked and counted the old way.
constexpr u64 AG_LARGE_BIN = 1u << 16;
constexpr u32 AGL_SLICE = 1u << 15;
constexpr int AGL_LOG2TAB = 11;
constexpr u32 AGL_TAB = 1u << AGL_LOG2TAB;
constexpr u32 AGL_TABLES = 16;
constexpr int AGL_MAX_PROBE = 64;
struct AggLarge {
    u32 *bin_tab;                  // [nbins] 0, or 1 + the bin's table
    unsigned long long *tkeys;     // [AGL_TABLES][AGL_TAB], AG_EMPTY (two-word keys: word 1)
    unsigned long long *tkeys0;    // two-word keys: word 0 (AG_EMPTY until the slot's claimer has stored it)
    u32 *tcnt;                     // [AGL_TABLES][AGL_TAB], 0
    u32 *tbad;                     // [AGL_TABLES] 1: the bin is counted the old way after all
    unsigned long long *units;     // {bin << 32 | slice}; room for n / AGL_SLICE + AGL_TABLES + 1
    u32 *ctl;                      // [0] tables handed out, [1] units listed, [2] ticket of the slice kernel
};
// the bin this workgroup works on (false: none) and what to do when it overflows
__device__ __forceinline__ bool agg_pick_bin(const AggTask &t, u32 nbins, u32 &b)
{
    b = blockIdx.x;
    if (t.bin_list) { if (b >= *t.bin_list_n || b >= nbins) return false; b = t.bin_list[b]; }
    return b < nbins;                                   // (a list entry is a bin id: anything else would be a bug upstream, never an address)
}
__device__ __forceinline__ void agg_bin_overflow(const AggTask &t, u32 nbins, u32 b, u32 cap)
{
    if (t.ovf_list) { const u32 at = atomicAdd(t.ovf_n, 1u); if (at < nbins) t.ovf_list[at] = b; else atomicOr(t.flags, (u32)AG_FLAG_OVERFLOW); }   // (a bin is listed once per rung: at < nbins)
    else atomicOr(t.flags, (u32)AG_FLAG_OVERFLOW);
    atomicMax(t.flags + AG_BATCH, cap);
    t.bin_cnt[b] = 0;
}
struct AggArgs { AggTask t[AG_BATCH]; u32 lower, upper; u32 nbins; int shift; int nw; int top_bits; int top_sig; };   // top_sig: significant bits of the most significant word (multi-word keys)
//   // bins = key >> shift, nbins of them (65536 / 48, or 256 / 56)
// top_bits (multi-word keys): how many of the 16 prefix bits the most significant word holds (0 or 16: all); the rest are the top
// bits of the word below
__device__ __forceinline__ u32 agg_bin_of(const AggArgs &a, const u64 *rec)
{
    if (a.top_bits > 0 && a.top_bits < 16)
        return ((u32)(rec[a.nw - 1] >> (64 - a.top_bits)) << (16 - a.top_bits)) | (u32)(rec[a.nw - 2] >> (48 + a.top_bits));
    return (u32)(rec[a.nw - 1] >> a.shift);
}

// bounds[b] = index of the first key whose top bits are >= b (b = 0 .. AG_BINS); one thread per bound
__global__ __launch_bounds__(AG_THREADS) void bin_bounds_kernel(AggArgs a)
{
    const AggTask &t = a.t[blockIdx.y];
    const u32 b = blockIdx.x * AG_THREADS + threadIdx.x;
    if (!t.active || b > a.nbins) return;
    u64 lo = 0, hi = t.n;                               // first index in [0, n] with (key >> shift) >= b
    if (b == a.nbins) lo = t.n;
    else while (lo < hi) {
        const u64 mid = lo + ((hi - lo) >> 1);
        if (agg_bin_of(a, t.keys + mid * a.nw) < b) lo = mid + 1; else hi = mid;    // (the prefix sits in the most significant word, or continues in the one below)
    }
    t.bounds[b] = lo;
}

template <int LOG2CAP>
__device__ __forceinline__ u32 agg_slot(u64 k)
{
    const u32 x = (u32)(k >> 32) ^ (u32)k;
    return (x * 0x9E3779B1u) >> (32 - LOG2CAP);
}

// Counts one key per lane of `act` (a lane mask) in the LDS table: the probe sequence of section 1 of agg_finish_kernel as a
// loop of the WAVE.  Per probe the slot of every lane still in `act` is read, an empty slot is claimed with the lane's key
// (compare-and-swap), lanes that found their key or claimed the slot add one to its counter and leave `act`; the others move
// to the next slot.  Returns the lanes that ran out of probes (0: all counted); h leaves as the slot the lane's key was counted in.
// Written in assembly because the scalar unit is what this kernel runs out of (8 more scalar instructions per probe cost
// 14 % of its time): the compiler spends ~25 scalar instructions per probe on execution-mask bookkeeping for the equivalent
// if / break structure, this loop 6 (9 when a slot is claimed, 4 more per further probe).
//   key_base / cnt_base: LDS byte addresses of the key and count arrays; h: first slot (CAP = MASK + 1 slots)
template <u32 MASK>
__device__ __forceinline__ u64 agg_count_keys(u64 act, u32 key_base, u32 cnt_base, u32 &h, u64 k, u32 inc = 1u)      // inc: what the key's counter grows by (1: an instance; a pair's count in the weighted finish)
{
    u64 save, t, cur;
    u32 ka, ca, p;
    const u64 empty = AG_EMPTY;
    const u32 one = inc;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_movk_i32 %[p], 1\n\t"
        "s_mov_b64 exec, %[act]\n"
        "0:\n\t"
        "v_lshl_add_u32 %[ka], %[h], 3, %[kb]\n\t"
        "ds_read_b64 %[cur], %[ka]\n\t"
        "v_lshl_add_u32 %[ca], %[h], 2, %[cb]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, -1, %[cur]\n\t"
        "s_cbranch_vccz 1f\n\t"                              // no empty slot among the lanes: nothing to claim
        "s_mov_b64 exec, vcc\n\t"
        "ds_cmpst_rtn_b64 %[cur], %[ka], %[emp], %[k]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 exec, %[act]\n\t"
        "v_cmp_eq_u64 vcc, -1, %[cur]\n\t"                   // claimed ...
        "v_cmp_eq_u64 %[t], %[cur], %[k]\n\t"                // ... or found
        "s_or_b64 vcc, vcc, %[t]\n\t"
        "s_branch 3f\n"
        "1:\n\t"
        "v_cmp_eq_u64 vcc, %[cur], %[k]\n"
        "3:\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_add_u32 %[ca], %[one]\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"               // (scc: lanes left)
        "s_cbranch_scc0 2f\n\t"
        "s_mov_b64 exec, %[act]\n\t"
        "v_add_u32 %[h], %[p], %[h]\n\t"                    // triangular steps (1, 2, 3, ...: every slot of a power-of-two table once): no clusters --
        "v_and_b32 %[h], %[mask], %[h]\n\t"                 // with steps of one a bin of 915 keys in 2048 slots ran past 32 probes one time in twenty
        "s_add_u32 %[p], %[p], 1\n\t"
        "s_cmp_lg_u32 %[p], %[maxp]\n\t"
        "s_cbranch_scc1 0b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[save]"
        : [act] "+s"(act), [h] "+v"(h), [save] "=&s"(save), [t] "=&s"(t), [p] "=&s"(p), [cur] "=&v"(cur), [ka] "=&v"(ka), [ca] "=&v"(ca)
        : [kb] "s"(key_base), [cb] "s"(cnt_base), [k] "v"(k), [emp] "v"(empty), [one] "v"(one), [mask] "n"(MASK), [maxp] "n"(AG_MAX_PROBE + 1)
        : "vcc", "scc", "memory");
    return act;
}

// (bins of very many records, AggLarge above: a wave first asks whether its lanes all hold the same key and, if so, sends ONE lane with the
//  sum -- same-address LDS atomics are taken one lane after the other)
__device__ __forceinline__ u32 wave_sum_u32(u32 v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE); return v; }

// Steps 2 and 3 of the aggregation for one bin whose records have been counted into the table {s_key, s_cnt} (CAP slots):
// compact the distinct keys, order them, filter [L, U], write the entries to the bin's slots, publish the count.
// Called by all threads of the workgroup (barriers inside).
template <int LOG2CAP>
__device__ __forceinline__ void agg_emit_bin(const AggArgs &a, const AggTask &t, u32 b, u64 s, u64 *s_key, u32 *s_cnt, u32 *s_bkt, u32 *s_scr)
{
    constexpr int CAP = 1 << LOG2CAP;
    constexpr int PER = CAP / AG_THREADS;
    constexpr int NBKT = CAP / 2;
    const int tid = threadIdx.x;
    // ---- 2. compact the occupied slots (in place: every lane holds its PER slots in registers across the
    //         barrier), sort the distinct keys ---------------------------------------------------------------
    // The frequency filter comes FIRST: only the keys that stay are ordered (reads with 1 % errors: ~900 distinct keys per bin, ~120 of them
    // within [L, U] -- ordering all of them and dropping seven in eight afterwards was 40 % of the kernel there, measured).  The slots hold
    // both numbers in one scan: occupied slots in the low half (the host's feedback), kept ones in the high half.
    u64 mk[PER]; u32 mc[PER];
    u32 occ = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        mk[j] = s_key[tid * PER + j]; mc[j] = s_cnt[tid * PER + j];
        const bool used = mk[j] != AG_EMPTY;
        if (!(used && mc[j] >= a.lower && mc[j] <= a.upper)) mk[j] = AG_EMPTY;
        occ += (used ? 1u : 0u) + (mk[j] != AG_EMPTY ? 0x10000u : 0u);
    }
    u32 D;
    u32 o = block_excl_scan_256<u32>(occ, s_scr, &D) >> 16;      // (two barriers inside: all slots are read before any is rewritten)
    D = (u32)__builtin_amdgcn_readfirstlane((int)D);        // (the same in every lane: loops and branches on it are the wave's, scalar)
    const u32 d_all = D & 0xFFFFu;
    D >>= 16;
#pragma unroll
    for (int j = 0; j < PER; ++j) if (mk[j] != AG_EMPTY) { s_key[o] = mk[j]; s_cnt[o] = mc[j]; ++o; }
    __syncthreads();
    if (tid == 0 && d_all > 256u) atomicMax(t.flags + AG_BATCH, d_all);      // (feedback for the host's choice of the first table; small bins are the common case and stay silent)
    if (D <= (u32)AG_THREADS) {
        // rank by counting: keys are distinct, so ranks are a permutation; s_key[j] is a broadcast read
        u64 k = 0; u32 c = 0, r = 0;
        if ((u32)(tid & ~(WAVE - 1)) < D) {                // (whole waves: the loop below is scalar, lanes past D rank an empty key and drop it)
            const bool mine = (u32)tid < D;
            k = mine ? s_key[tid] : AG_EMPTY; c = mine ? s_cnt[tid] : 0;
            u32 j = 0;
            for (; j + 4 <= D; j += 4) {                   // four broadcast reads per step
                const u64 k0 = s_key[j], k1 = s_key[j + 1], k2 = s_key[j + 2], k3 = s_key[j + 3];
                r += (k0 < k ? 1u : 0u) + (k1 < k ? 1u : 0u) + (k2 < k ? 1u : 0u) + (k3 < k ? 1u : 0u);
            }
            for (; j < D; ++j) r += s_key[j] < k;
        }
        __syncthreads();
        if ((u32)tid < D) { s_key[r] = k; s_cnt[r] = c; }
        __syncthreads();
    } else {
        // Hundreds to thousands of distinct keys (low coverage, reads with errors): counting sort on the key bits right below
        // the bin prefix -- the keys of a bin are close to uniform there, about one per bucket --, then every key ranks itself
        // inside its bucket: ~10 LDS operations per key instead of the ~log^2 of a sorting network; a bucket that collects
        // many keys (a shared prefix) costs only its own square.  In place: every lane keeps its keys in registers across
        // the barriers between "all read" and "all write".
        constexpr int EPT = CAP / AG_THREADS;
        constexpr int LOG2BKT = LOG2CAP - 1;
        const int bshift = a.shift - LOG2BKT;              // (the prefix bits are equal inside a bin: masked off below)
        for (int i = tid; i < NBKT; i += AG_THREADS) s_bkt[i] = 0;
        __syncthreads();
        u64 ek[EPT]; u32 ec[EPT], er[EPT];
#pragma unroll
        for (int x = 0; x < EPT; ++x) {
            const u32 i = x * AG_THREADS + tid;
            ek[x] = AG_EMPTY; ec[x] = 0; er[x] = 0;
            if (i < D) { ek[x] = s_key[i]; ec[x] = s_cnt[i]; er[x] = atomicAdd(&s_bkt[(u32)(ek[x] >> bshift) & (NBKT - 1)], 1u); }
        }
        __syncthreads();
        {
            constexpr int BPT = NBKT / AG_THREADS;
            u32 v[BPT], sum = 0;
#pragma unroll
            for (int x = 0; x < BPT; ++x) { v[x] = s_bkt[tid * BPT + x]; sum += v[x]; }
            u32 ex = block_excl_scan_256<u32>(sum, s_scr, nullptr);
#pragma unroll
            for (int x = 0; x < BPT; ++x) { s_bkt[tid * BPT + x] = ex; ex += v[x]; }
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < EPT; ++x)
            if (ek[x] != AG_EMPTY) s_key[s_bkt[(u32)(ek[x] >> bshift) & (NBKT - 1)] + er[x]] = ek[x];        // bucket-major (counts follow below)
        __syncthreads();
#pragma unroll
        for (int x = 0; x < EPT; ++x) {
            if (ek[x] == AG_EMPTY) continue;
            const u32 bk = (u32)(ek[x] >> bshift) & (NBKT - 1);
            const u32 b0 = s_bkt[bk], b1 = (bk + 1 < (u32)NBKT) ? s_bkt[bk + 1] : D;
            u32 r = b0;
            for (u32 q = b0; q < b1; ++q) r += s_key[q] < ek[x];
            er[x] = r;
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < EPT; ++x) if (ek[x] != AG_EMPTY) { s_key[er[x]] = ek[x]; s_cnt[er[x]] = ec[x]; }
        __syncthreads();
    }
    const u64 *sk = s_key; const u32 *sc = s_cnt;

    // ---- 3. the entries in key order to the bin's slots (16-byte stores from consecutive lanes) ------------------
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(t.scratch + (s >> t.slot_shift) * 2);
    for (u32 i = tid; i < D; i += AG_THREADS) dst[i] = make_ulonglong2(sk[i], (u64)sc[i]);
    if (tid == 0) t.bin_cnt[b] = D;
}

// Diagnostic build only (-DHSK_DIAG): shader-clock sums per phase of agg_finish_kernel, stamped by thread 0 of every workgroup
#ifdef HSK_DIAG
__device__ unsigned long long g_agg_diag[16];
#define AG_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ag_acc[i] = t_ - ag_last; ag_last = t_; } } while (0)
#else
#define AG_STAMP(i) do { } while (0)
#endif

// one thread per bin: a bin of AG_LARGE_BIN records and more gets a global table and its slices are listed
__global__ __launch_bounds__(AG_THREADS) void agg_large_list_kernel(AggArgs a)
{
    const AggTask &t = a.t[blockIdx.y];
    const u32 b = blockIdx.x * AG_THREADS + threadIdx.x;
    if (!t.active || !t.lg || b >= a.nbins) return;
    const u64 n = t.bounds[b + 1] - t.bounds[b];
    if (n < AG_LARGE_BIN) return;
    const AggLarge &lg = *t.lg;
    const u32 ti = atomicAdd(&lg.ctl[0], 1u);
    if (ti >= AGL_TABLES) return;
    const u32 ns = (u32)((n + AGL_SLICE - 1) / AGL_SLICE);
    const u32 u0 = atomicAdd(&lg.ctl[1], ns);           // (fits: the listed bins' slices are at most n / AGL_SLICE + AGL_TABLES)
    for (u32 j = 0; j < ns; ++j) lg.units[u0 + j] = ((unsigned long long)b << 32) | j;
    lg.bin_tab[b] = ti + 1u;
}

// persistent workgroups, one slice per ticket: the slice's records into an LDS table, the table's pairs into the bin's global table
__global__ __launch_bounds__(AG_THREADS) void agg_large_slice_kernel(AggArgs a)
{
    constexpr int LOG2CAP = AG_LOG2CAP_MEDIUM, CAP = 1 << LOG2CAP, PER = CAP / AG_THREADS, UNR = 16;
    __shared__ u64 s_key[CAP];
    __shared__ u32 s_cnt[CAP];
    __shared__ u32 s_ctl[2];
    const AggTask &t = a.t[blockIdx.y];
    if (!t.active || !t.lg) return;
    const AggLarge &lg = *t.lg;
    const u32 nunits = lg.ctl[1];
    if (nunits == 0) return;
    const int tid = threadIdx.x;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const u32 key_lds = (u32)(uintptr_t)(LdsPtr)s_key, cnt_lds = (u32)(uintptr_t)(LdsPtr)s_cnt;
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_ctl[0] = atomicAdd(&lg.ctl[2], 1u); s_ctl[1] = 0; }
#pragma unroll
        for (int j = 0; j < PER; ++j) { s_key[j * AG_THREADS + tid] = AG_EMPTY; s_cnt[j * AG_THREADS + tid] = 0; }
        __syncthreads();
        const u32 u = s_ctl[0];
        if (u >= nunits) break;
        const unsigned long long un = lg.units[u];
        const u32 b = (u32)(un >> 32), ti = lg.bin_tab[b] - 1u;
        const u64 s = t.bounds[b] + (u64)(u32)un * AGL_SLICE, be = t.bounds[b + 1], e = s + AGL_SLICE < be ? s + AGL_SLICE : be;
        for (u64 i = s + tid; i < e; i += (u64)AG_THREADS * UNR) {
            u64 k[UNR];
#pragma unroll
            for (int x = 0; x < UNR; ++x) { const u64 idx = i + (u64)x * AG_THREADS; k[x] = idx < e ? t.keys[idx] : AG_EMPTY; }
#pragma unroll
            for (int x = 0; x < UNR; ++x) {
                u64 act = __ballot(k[x] != AG_EMPTY);
                if (act == 0) continue;                   // (uniform; the active lanes are a prefix of the wave: lane 0 is one of them)
                u32 h = agg_slot<LOG2CAP>(k[x]), inc = 1u;
                const u64 k0 = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(k[x] >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)k[x]);
                if (__ballot(k[x] == k0) == act) { inc = (u32)__popcll(act); act = 1ULL; }
                if (agg_count_keys<(u32)CAP - 1u>(act, key_lds, cnt_lds, h, k[x], inc) != 0) s_ctl[1] = 1;      // (uniform)
            }
            if (__hip_atomic_load(&s_ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
        __syncthreads();
        if (s_ctl[1]) { if (tid == 0) lg.tbad[ti] = 1u; continue; }      // (more distinct keys than a slice's table takes: not this kind of bin)
        unsigned long long *gk = lg.tkeys + (u64)ti * AGL_TAB; u32 *gc = lg.tcnt + (u64)ti * AGL_TAB;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const u64 kq = s_key[j * AG_THREADS + tid];
            if (kq == AG_EMPTY) continue;
            const u32 cq = s_cnt[j * AG_THREADS + tid];
            u32 h = agg_slot<AGL_LOG2TAB>(kq);
            bool placed = false;
            for (int pr = 1; pr <= AGL_MAX_PROBE && !placed; ++pr) {
                unsigned long long prev = gk[h];
                if (prev == AG_EMPTY) prev = atomicCAS(&gk[h], (unsigned long long)AG_EMPTY, (unsigned long long)kq);
                if (prev == AG_EMPTY || prev == kq) { atomicAdd(&gc[h], cq); placed = true; }
                else h = (h + (u32)pr) & (AGL_TAB - 1u);
            }
            if (!placed) lg.tbad[ti] = 1u;
        }
    }
}

// W: the records are {key, count} pairs (AggTask::vals) and a key's counter grows by the pair's count
template <int LOG2CAP, bool W = false>
__global__ __launch_bounds__(AG_THREADS) void agg_finish_kernel(AggArgs a)
{
#ifdef HSK_DIAG
    unsigned long long ag_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ag_last = __builtin_amdgcn_s_memtime();
#endif
    constexpr int CAP = 1 << LOG2CAP;
    constexpr int PER = CAP / AG_THREADS;
    constexpr int NBKT = CAP / 2;   // buckets of the in-LDS counting sort that orders many distinct keys (one per slot of the idle half)
    __shared__ u64 s_key[CAP];      // hash table, then the distinct keys compacted, then sorted
    __shared__ u32 s_cnt[CAP];
    __shared__ u32 s_bkt[NBKT];
    __shared__ u32 s_scr[8];
    __shared__ u32 s_ovf;
    const AggTask &t = a.t[blockIdx.y];
    if (!t.active) return;
    u32 b;
    if (!agg_pick_bin(t, a.nbins, b)) return;
    const int tid = threadIdx.x;
    const u64 s = t.bounds[b], e = t.bounds[b + 1];
    if (e == s) { if (tid == 0) t.bin_cnt[b] = 0; return; }
    AG_STAMP(0);                                        // launch -> bounds known

#pragma unroll
    for (int j = 0; j < PER; ++j) { s_key[j * AG_THREADS + tid] = AG_EMPTY; s_cnt[j * AG_THREADS + tid] = 0; }
    if (tid == 0) s_ovf = 0;
    __syncthreads();
    AG_STAMP(1);                                        // table cleared

    // ---- 1. count the records of the bin in the table ------------------------------------------------------
    // 16 loads per lane are issued before the first insert: a typical bin (4096 records) costs one memory latency
    constexpr int AG_UNROLL = 16;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const u32 key_lds = (u32)(uintptr_t)(LdsPtr)s_key, cnt_lds = (u32)(uintptr_t)(LdsPtr)s_cnt;     // LDS byte addresses of the two arrays
    const bool large = e - s >= AG_LARGE_BIN;
    u32 ltab = 0;                                       // 1 + the global table the bin's slices were counted into (agg_large_slice_kernel)
    if (!W && large && t.lg) { ltab = t.lg->bin_tab[b]; if (ltab && t.lg->tbad[ltab - 1]) ltab = 0; }
    if (ltab) {
        const unsigned long long *gk = t.lg->tkeys + (u64)(ltab - 1) * AGL_TAB; const u32 *gc = t.lg->tcnt + (u64)(ltab - 1) * AGL_TAB;
        for (u32 q = tid; q < AGL_TAB; q += AG_THREADS) {
            const u64 kq = gk[q];
            const u64 act = __ballot(kq != AG_EMPTY);
            u32 h = agg_slot<LOG2CAP>(kq);
            if (act != 0 && agg_count_keys<(u32)CAP - 1u>(act, key_lds, cnt_lds, h, kq, gc[q]) != 0) s_ovf = 1;      // (uniform)
        }
    } else
    for (u64 i = s + tid; i < e; i += (u64)AG_THREADS * AG_UNROLL) {
        u64 k[AG_UNROLL]; u32 wv[W ? AG_UNROLL : 1];
#pragma unroll
        for (int u = 0; u < AG_UNROLL; ++u) {
            const u64 idx = i + (u64)u * AG_THREADS; k[u] = idx < e ? t.keys[idx] : AG_EMPTY;
            if (W) wv[u] = idx < e ? (u32)t.vals[idx] : 0u;
        }
#pragma unroll
        for (int u = 0; u < AG_UNROLL; ++u) {
            u64 act = __ballot(k[u] != AG_EMPTY);
            u32 h = agg_slot<LOG2CAP>(k[u]);
            u32 inc = W ? wv[u] : 1u;
            if (large && act != 0) {                      // (uniform; the active lanes are a prefix of the wave: lane 0 is one of them)
                const u64 k0 = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(k[u] >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)k[u]);
                if (__ballot(k[u] == k0) == act) { inc = W ? wave_sum_u32(k[u] != AG_EMPTY ? wv[u] : 0u) : (u32)__popcll(act); act = 1ULL; }
            }
            if (act != 0 && agg_count_keys<(u32)CAP - 1u>(act, key_lds, cnt_lds, h, k[u], inc) != 0) s_ovf = 1;      // (uniform)
        }
        // a bin with more distinct keys than the table takes (a probe sequence ran past AG_MAX_PROBE slots: with linear probing
        // that starts at a load of ~0.8) gives up here instead of grinding through the rest of its records
        if (__hip_atomic_load(&s_ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
    }
    __syncthreads();
    AG_STAMP(2);                                        // records loaded and counted
    if (s_ovf) {
        if (tid == 0) agg_bin_overflow(t, a.nbins, b, (u32)CAP);
        return;
    }

    // ---- 2, 3. distinct keys compacted, ordered, filtered, written -------------------------------------------
    agg_emit_bin<LOG2CAP>(a, t, b, s, s_key, s_cnt, s_bkt, s_scr);
    AG_STAMP(5);                                        // filtered and written
#ifdef HSK_DIAG
    if (tid == 0) { for (int i = 0; i < 6; ++i) atomicAdd(&g_agg_diag[i], ag_acc[i]); atomicAdd(&g_agg_diag[8], 1ULL); atomicAdd(&g_agg_diag[9], (unsigned long long)(e - s)); }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Two-word keys (32 < K < 64): the same aggregation over 16-bit prefix bins of the MOST significant word (word 1:
// the keys are ordered as little-endian two-word integers, hsk_sort.h).  There is no 128-bit LDS compare-and-swap,
// so a slot is claimed with a CAS on word 1 and the claimer then publishes word 0; a lane that finds its word 1 in
// a slot waits until word 0 is there (the claimer never waits for anything: claimers of the same wave have already
// issued their store, LDS operations of a wave execute in order) and compares it.  The "not yet published" marker
// ~0 is no valid word 0: a canonical k-mer that starts with 32 T needs a reverse complement that starts with 32 T
// too, i.e. a k-mer whose last 32 bases are A, impossible for K < 64 next to 32 leading T.
// ------------------------------------------------------------------------------------------------------------
// Many distinct keys (more than a workgroup has lanes) of NW words in key order, as agg_emit_bin does it for one word: a counting
// sort on the key bits right below the 16-bit bin prefix, then every key ranks itself inside its bucket (~10 LDS operations per
// key; the bitonic network this replaces for multi-word keys and EXTENSION made bins of ~1000 keys -- reads with 1 % errors --
// ten times slower).  The bits come from the significant bit string of the key, most significant word first: the top word holds
// top_sig of them (left-aligned, zeros below), the word under it continues.  kw[w]: word w of key i at kw[w][i]; s_slot (SLOT):
// carried along.  In place: every lane holds its keys in registers between "all read" and "all written".
template <int LOG2CAP, int NW, bool SLOT>
__device__ __forceinline__ void agg_order_many(u32 D, u64 *const (&kw)[NW], u32 *s_cnt, u16 *s_slot, u32 *s_bkt, u32 *s_scr, int top_sig)
{
    constexpr int CAP = 1 << LOG2CAP, EPT = CAP / AG_THREADS, NBKT = CAP / 2, LOG2BKT = LOG2CAP - 1;
    const int tid = threadIdx.x;
    auto bucket = [&](const u64 (&k)[NW]) -> u32 {
        u64 hi = k[NW - 1];
        if (NW > 1 && top_sig < 64) hi |= k[NW > 1 ? NW - 2 : 0] >> top_sig;
        return (u32)(hi >> (48 - LOG2BKT)) & (u32)(NBKT - 1);
    };
    for (int i = tid; i < NBKT; i += AG_THREADS) s_bkt[i] = 0;
    __syncthreads();
    u64 ek[EPT][NW]; u32 ec[EPT], er[EPT], eb[EPT]; u16 es[EPT];
#pragma unroll
    for (int x = 0; x < EPT; ++x) {
        const u32 i = x * AG_THREADS + tid;
        ec[x] = 0; er[x] = 0; eb[x] = ~0u; es[x] = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) ek[x][w] = 0;
        if (i < D) {
#pragma unroll
            for (int w = 0; w < NW; ++w) ek[x][w] = kw[w][i];
            ec[x] = s_cnt[i]; if (SLOT) es[x] = s_slot[i];
            eb[x] = bucket(ek[x]);
            er[x] = atomicAdd(&s_bkt[eb[x]], 1u);
        }
    }
    __syncthreads();
    {
        constexpr int BPT = NBKT / AG_THREADS;
        u32 v[BPT], sum = 0;
#pragma unroll
        for (int x = 0; x < BPT; ++x) { v[x] = s_bkt[tid * BPT + x]; sum += v[x]; }
        u32 ex = block_excl_scan_256<u32>(sum, s_scr, nullptr);
#pragma unroll
        for (int x = 0; x < BPT; ++x) { s_bkt[tid * BPT + x] = ex; ex += v[x]; }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < EPT; ++x)
        if (eb[x] != ~0u) {
            const u32 pos = s_bkt[eb[x]] + er[x];               // bucket-major
#pragma unroll
            for (int w = 0; w < NW; ++w) kw[w][pos] = ek[x][w];
        }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < EPT; ++x) {
        if (eb[x] == ~0u) continue;
        const u32 b0 = s_bkt[eb[x]], b1 = (eb[x] + 1 < (u32)NBKT) ? s_bkt[eb[x] + 1] : D;
        u32 r = b0;
        for (u32 q = b0; q < b1; ++q) {
            bool less = false, decided = false;
#pragma unroll
            for (int w = NW - 1; w >= 0; --w) { const u64 o = kw[w][q]; if (!decided && o != ek[x][w]) { less = o < ek[x][w]; decided = true; } }
            r += less ? 1u : 0u;
        }
        er[x] = r;
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < EPT; ++x)
        if (eb[x] != ~0u) {
#pragma unroll
            for (int w = 0; w < NW; ++w) kw[w][er[x]] = ek[x][w];
            s_cnt[er[x]] = ec[x]; if (SLOT) s_slot[er[x]] = es[x];
        }
    __syncthreads();
}

// agg_count_keys for two-word keys: a slot is claimed on word 1 (compare-and-swap), the claimer publishes word 0 and counts
// itself; a lane that finds its word 1 in the slot reads word 0 until it is there (claimers of the same wave have issued
// their store before -- the LDS executes a wave's operations in order --, claimers of other waves never wait for anything
// between the claim and the store) and counts itself if it is its own.  Returns the lanes that ran out of probes; timed_out:
// a word 0 never appeared (cannot happen; the bin is then redone on the next rung like an overflowing one).
template <u32 MASK>
__device__ __forceinline__ u64 agg2_count_keys(u64 act, u32 k1_base, u32 k0_base, u32 cnt_base, u32 &h, u64 w1, u64 w0, u32 &timed_out, u32 inc = 1u)      // inc: as in agg_count_keys
{
    u64 save, cur, v;
    u32 ka1, ka0, ca, p, spin, tmo = 0;
    const u64 empty = AG_EMPTY;
    const u32 one = inc;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_movk_i32 %[p], 1\n\t"
        "s_mov_b64 exec, %[act]\n"
        "0:\n\t"
        "v_lshl_add_u32 %[ka1], %[h], 3, %[k1b]\n\t"
        "ds_read_b64 %[cur], %[ka1]\n\t"
        "v_lshl_add_u32 %[ka0], %[h], 3, %[k0b]\n\t"
        "v_lshl_add_u32 %[ca], %[h], 2, %[cb]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, -1, %[cur]\n\t"
        "s_cbranch_vccz 1f\n\t"                               // no empty slot among the lanes
        "s_mov_b64 exec, vcc\n\t"
        "ds_cmpst_rtn_b64 %[cur], %[ka1], %[emp], %[w1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, -1, %[cur]\n\t"                    // claimed: publish word 0, count
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b64 %[ka0], %[w0]\n\t"
        "ds_add_u32 %[ca], %[one]\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"
        "s_cbranch_scc0 2f\n\t"
        "s_mov_b64 exec, %[act]\n"
        "1:\n\t"
        "v_cmp_eq_u64 vcc, %[cur], %[w1]\n\t"                 // word 1 is there: is word 0 mine?
        "s_cbranch_vccz 4f\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "s_movk_i32 %[spin], 0x7fff\n"
        "5:\n\t"
        "ds_read_b64 %[v], %[ka0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, -1, %[v]\n\t"                      // not published yet
        "s_cbranch_vccz 6f\n\t"
        "s_sub_u32 %[spin], %[spin], 1\n\t"
        "s_cmp_lg_u32 %[spin], 0\n\t"
        "s_cbranch_scc1 5b\n\t"
        "s_mov_b32 %[tmo], 1\n"
        "6:\n\t"
        "v_cmp_eq_u64 vcc, %[v], %[w0]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_add_u32 %[ca], %[one]\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"
        "s_cbranch_scc0 2f\n\t"
        "s_mov_b64 exec, %[act]\n"
        "4:\n\t"
        "v_add_u32 %[h], %[p], %[h]\n\t"                    // (triangular steps, as agg_count_keys)
        "v_and_b32 %[h], %[mask], %[h]\n\t"
        "s_add_u32 %[p], %[p], 1\n\t"
        "s_cmp_lg_u32 %[p], %[maxp]\n\t"
        "s_cbranch_scc1 0b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[save]"
        : [act] "+s"(act), [h] "+v"(h), [tmo] "+s"(tmo), [save] "=&s"(save), [p] "=&s"(p), [spin] "=&s"(spin),
          [cur] "=&v"(cur), [v] "=&v"(v), [ka1] "=&v"(ka1), [ka0] "=&v"(ka0), [ca] "=&v"(ca)
        : [k1b] "s"(k1_base), [k0b] "s"(k0_base), [cb] "s"(cnt_base), [w1] "v"(w1), [w0] "v"(w0), [emp] "v"(empty), [one] "v"(one),
          [mask] "n"(MASK), [maxp] "n"(AG_MAX_PROBE + 1)
        : "vcc", "scc", "memory");
    timed_out |= tmo;
    return act;
}

__device__ __forceinline__ bool key2_less(u64 a1, u64 a0, u64 b1, u64 b0) { return a1 < b1 || (a1 == b1 && a0 < b0); }

// the same for two-word keys: the slice's table is filled by agg2_count_keys; in the bin's global table a slot is claimed on word 1 and word 0
// published behind it, a workgroup that finds its word 1 waits for the word 0 (the claimer stores it right after its claim)
__global__ __launch_bounds__(AG_THREADS) void agg2_large_slice_kernel(AggArgs a)
{
    constexpr int LOG2CAP = AG_LOG2CAP_SMALL, CAP = 1 << LOG2CAP, PER = CAP / AG_THREADS, UNR = 8;
    __shared__ u64 s_k1[CAP];
    __shared__ u64 s_k0[CAP];
    __shared__ u32 s_cnt[CAP];
    __shared__ u32 s_ctl[2];
    const AggTask &t = a.t[blockIdx.y];
    if (!t.active || !t.lg) return;
    const AggLarge &lg = *t.lg;
    const u32 nunits = lg.ctl[1];
    if (nunits == 0) return;
    const int tid = threadIdx.x;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const u32 k1_lds = (u32)(uintptr_t)(LdsPtr)s_k1, k0_lds = (u32)(uintptr_t)(LdsPtr)s_k0, cnt_lds = (u32)(uintptr_t)(LdsPtr)s_cnt;
    const ulonglong2 *recs = reinterpret_cast<const ulonglong2 *>(t.keys);       // {word 0, word 1}
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_ctl[0] = atomicAdd(&lg.ctl[2], 1u); s_ctl[1] = 0; }
#pragma unroll
        for (int j = 0; j < PER; ++j) { s_k1[j * AG_THREADS + tid] = AG_EMPTY; s_k0[j * AG_THREADS + tid] = AG_EMPTY; s_cnt[j * AG_THREADS + tid] = 0; }
        __syncthreads();
        const u32 u = s_ctl[0];
        if (u >= nunits) break;
        const unsigned long long un = lg.units[u];
        const u32 b = (u32)(un >> 32), ti = lg.bin_tab[b] - 1u;
        const u64 s = t.bounds[b] + (u64)(u32)un * AGL_SLICE, be = t.bounds[b + 1], e = s + AGL_SLICE < be ? s + AGL_SLICE : be;
        for (u64 i = s + tid; i < e; i += (u64)AG_THREADS * UNR) {
            ulonglong2 k[UNR];
#pragma unroll
            for (int x = 0; x < UNR; ++x) { const u64 idx = i + (u64)x * AG_THREADS; k[x] = idx < e ? recs[idx] : make_ulonglong2(AG_EMPTY, AG_EMPTY); }
#pragma unroll
            for (int x = 0; x < UNR; ++x) {
                const u64 w0 = k[x].x, w1 = k[x].y;
                u64 act = __ballot(w1 != AG_EMPTY);
                if (act == 0) continue;                   // (uniform)
                const u64 m = w0 ^ (w1 >> 9) ^ (w1 << 21);
                u32 tmo = 0, h = (((u32)(m >> 32) ^ (u32)m) * 0x9E3779B1u) >> (32 - LOG2CAP), inc = 1u;
                const u64 f1 = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(w1 >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)w1);
                const u64 f0 = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(w0 >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)w0);
                if (__ballot(w1 == f1 && w0 == f0) == act) { inc = (u32)__popcll(act); act = 1ULL; }
                if (agg2_count_keys<(u32)CAP - 1u>(act, k1_lds, k0_lds, cnt_lds, h, w1, w0, tmo, inc) != 0 || tmo) s_ctl[1] = 1;      // (uniform)
            }
            if (__hip_atomic_load(&s_ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
        __syncthreads();
        if (s_ctl[1]) { if (tid == 0) lg.tbad[ti] = 1u; continue; }
        unsigned long long *g1 = lg.tkeys + (u64)ti * AGL_TAB, *g0 = lg.tkeys0 + (u64)ti * AGL_TAB; u32 *gc = lg.tcnt + (u64)ti * AGL_TAB;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const u64 q1 = s_k1[j * AG_THREADS + tid], q0 = s_k0[j * AG_THREADS + tid];
            if (q1 == AG_EMPTY) continue;
            const u32 cq = s_cnt[j * AG_THREADS + tid];
            const u64 m = q0 ^ (q1 >> 9) ^ (q1 << 21);
            u32 h = (((u32)(m >> 32) ^ (u32)m) * 0x9E3779B1u) >> (32 - AGL_LOG2TAB);
            bool placed = false;
            for (int pr = 1; pr <= AGL_MAX_PROBE && !placed; ++pr) {
                unsigned long long prev = __hip_atomic_load(&g1[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev == AG_EMPTY) {
                    prev = atomicCAS(&g1[h], (unsigned long long)AG_EMPTY, (unsigned long long)q1);
                    if (prev == AG_EMPTY) { __hip_atomic_store(&g0[h], (unsigned long long)q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(&gc[h], cq); placed = true; break; }
                }
                if (prev == q1) {
                    unsigned long long v0 = AG_EMPTY; u32 spins = 0;
                    while ((v0 = __hip_atomic_load(&g0[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == AG_EMPTY && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
                    if (v0 == q0) { atomicAdd(&gc[h], cq); placed = true; break; }
                    if (v0 == AG_EMPTY) break;            // (a word 0 that never came: cannot happen; the bin is counted the old way)
                }
                h = (h + (u32)pr) & (AGL_TAB - 1u);
            }
            if (!placed) lg.tbad[ti] = 1u;
        }
    }
}

template <int LOG2CAP, bool W = false>      // W: the records are {k-mer, count} pairs (AggTask::vals), the table adds the counts up (combining extraction, two-word keys)
__global__ __launch_bounds__(AG_THREADS) void agg2_finish_kernel(AggArgs a)
{
    constexpr int CAP = 1 << LOG2CAP;
    constexpr int PER = CAP / AG_THREADS;
    __shared__ u64 s_k1[CAP];
    __shared__ u64 s_k0[CAP];
    __shared__ u32 s_cnt[CAP];
    __shared__ u32 s_bkt[CAP / 2];  // buckets of the counting sort that orders many distinct keys
    __shared__ u32 s_scr[8];
    __shared__ u32 s_ovf;
    const AggTask &t = a.t[blockIdx.y];
    if (!t.active) return;
    u32 b;
    if (!agg_pick_bin(t, a.nbins, b)) return;
    const int tid = threadIdx.x;
    const u64 s = t.bounds[b], e = t.bounds[b + 1];
    if (e == s) { if (tid == 0) t.bin_cnt[b] = 0; return; }
#pragma unroll
    for (int j = 0; j < PER; ++j) { s_k1[j * AG_THREADS + tid] = AG_EMPTY; s_k0[j * AG_THREADS + tid] = AG_EMPTY; s_cnt[j * AG_THREADS + tid] = 0; }
    if (tid == 0) s_ovf = 0;
    __syncthreads();

    constexpr int UNR = 8;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const u32 k1_lds = (u32)(uintptr_t)(LdsPtr)s_k1, k0_lds = (u32)(uintptr_t)(LdsPtr)s_k0, cnt_lds = (u32)(uintptr_t)(LdsPtr)s_cnt;
    const bool large = e - s >= AG_LARGE_BIN;
    const ulonglong2 *recs = reinterpret_cast<const ulonglong2 *>(t.keys);       // {word 0, word 1}
    u32 ltab = 0;                                       // 1 + the global table the bin's slices were counted into (agg2_large_slice_kernel)
    if (!W && large && t.lg) { ltab = t.lg->bin_tab[b]; if (ltab && t.lg->tbad[ltab - 1]) ltab = 0; }
    if (ltab) {
        const unsigned long long *g1 = t.lg->tkeys + (u64)(ltab - 1) * AGL_TAB, *g0 = t.lg->tkeys0 + (u64)(ltab - 1) * AGL_TAB; const u32 *gc = t.lg->tcnt + (u64)(ltab - 1) * AGL_TAB;
        for (u32 q = tid; q < AGL_TAB; q += AG_THREADS) {
            const u64 w1 = g1[q], w0 = g0[q];
            const u64 act = __ballot(w1 != AG_EMPTY);
            if (act == 0) continue;                       // (uniform)
            const u64 m = w0 ^ (w1 >> 9) ^ (w1 << 21);
            u32 tmo = 0, h = (((u32)(m >> 32) ^ (u32)m) * 0x9E3779B1u) >> (32 - LOG2CAP);
            if (agg2_count_keys<(u32)CAP - 1u>(act, k1_lds, k0_lds, cnt_lds, h, w1, w0, tmo, gc[q]) != 0 || tmo) s_ovf = 1;      // (uniform)
        }
    } else
    for (u64 i = s + tid; i < e; i += (u64)AG_THREADS * UNR) {
        ulonglong2 k[UNR]; u32 wt[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { const u64 idx = i + (u64)u * AG_THREADS; k[u] = idx < e ? recs[idx] : make_ulonglong2(AG_EMPTY, AG_EMPTY); wt[u] = (W && idx < e) ? (u32)t.vals[idx] : 1u; }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const u64 w0 = k[u].x, w1 = k[u].y;
            u64 act = __ballot(w1 != AG_EMPTY);
            if (act == 0) continue;                       // (uniform)
            const u64 m = w0 ^ (w1 >> 9) ^ (w1 << 21);
            const u32 x = (u32)(m >> 32) ^ (u32)m;
            u32 tmo = 0, h = (x * 0x9E3779B1u) >> (32 - LOG2CAP), inc = wt[u];
            if (large) {                                  // (uniform; one k-mer in all lanes of the wave: one lane with the sum, as in agg_finish_kernel)
                const u64 f1 = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(w1 >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)w1);
                const u64 f0 = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(w0 >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)w0);
                if (__ballot(w1 == f1 && w0 == f0) == act) { inc = wave_sum_u32(w1 != AG_EMPTY ? wt[u] : 0u); act = 1ULL; }
            }
            if (agg2_count_keys<(u32)CAP - 1u>(act, k1_lds, k0_lds, cnt_lds, h, w1, w0, tmo, inc) != 0 || tmo) s_ovf = 1;      // (uniform)
        }
        if (__hip_atomic_load(&s_ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
    }
    __syncthreads();
    if (s_ovf) {
        if (tid == 0) agg_bin_overflow(t, a.nbins, b, (u32)CAP);
        return;
    }

    // ---- compact, order by (word 1, word 0) ----------------------------------------------------------------------
    u32 D;
    {
        u64 m1[PER], m0[PER]; u32 mc[PER];
        u32 occ = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { m1[j] = s_k1[tid * PER + j]; m0[j] = s_k0[tid * PER + j]; mc[j] = s_cnt[tid * PER + j]; occ += m1[j] != AG_EMPTY; }
        u32 o = block_excl_scan_256<u32>(occ, s_scr, &D);
#pragma unroll
        for (int j = 0; j < PER; ++j) if (m1[j] != AG_EMPTY) { s_k1[o] = m1[j]; s_k0[o] = m0[j]; s_cnt[o] = mc[j]; ++o; }
    }
    __syncthreads();
    if (D <= (u32)AG_THREADS) {
        u64 k1 = 0, k0 = 0; u32 c = 0, r = 0;
        if ((u32)tid < D) {
            k1 = s_k1[tid]; k0 = s_k0[tid]; c = s_cnt[tid];
            for (u32 j = 0; j < D; ++j) r += key2_less(s_k1[j], s_k0[j], k1, k0);
        }
        __syncthreads();
        if ((u32)tid < D) { s_k1[r] = k1; s_k0[r] = k0; s_cnt[r] = c; }
        __syncthreads();
    } else {
        u64 *const kw[2] = {s_k0, s_k1};
        agg_order_many<LOG2CAP, 2, false>(D, kw, s_cnt, nullptr, s_bkt, s_scr, a.top_sig);
    }

    // ---- filter, entries {word 0, word 1, count} in key order to the bin's slots -----------------------------------
    u32 kept = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const u32 i = tid * PER + j;
        if (i < D) { const u32 c = s_cnt[i]; kept += (c >= a.lower && c <= a.upper); }
    }
    u32 tot;
    const u32 w = block_excl_scan_256<u32>(kept, s_scr, &tot);
    u64 *dst = t.scratch + ((s >> t.slot_shift) + w) * 3;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const u32 i = tid * PER + j;
        if (i < D) {
            const u32 c = s_cnt[i];
            if (c >= a.lower && c <= a.upper) { dst[0] = s_k0[i]; dst[1] = s_k1[i]; dst[2] = (u64)c; dst += 3; }
        }
    }
    if (tid == 0) t.bin_cnt[b] = tot;
}

// ------------------------------------------------------------------------------------------------------------
// Three-word keys (64 < K <= 95): the same plan as for two
// words -- a slot is claimed with a compare-and-swap on the most significant word, then words 1 and 0 are stored --, but
// "published" cannot be a marker value of word 0 any more (a canonical k-mer of 64 bases and more can begin with 32 T and
// end with 32 A, so every value of word 0 occurs): the slot's COUNT is the flag.  The claimer stores the two words and
// then counts itself (0 -> 1); a lane that finds its word 2 in a slot waits until the count is non-zero (the claimer never
// waits for anything in between; the LDS executes a wave's operations in order) and compares words 1 and 0.  Word 2 holds
// K - 64 <= 31 bases left-aligned, so ~0 is never a word 2 and marks the empty slot.  Records and entries: {word 0, word 1,
// word 2[, count]}.  Before this kernel existed these keys took 20 full LSD passes over 24-byte records (K=77: 580 of 740 ms).
// ------------------------------------------------------------------------------------------------------------
// The probe loop of three-word keys (protocol above), a loop of the wave like agg_count_keys / agg2_count_keys.  In assembly
// for the scalar unit's sake and because the ORDER matters: within one wave the claimers must have stored and counted before
// the lanes that found their word 2 start to wait -- written as two `if` blocks the compiler is free to run the waiting lanes
// first (it did: every waiter timed out).  Returns the lanes that ran out of probes; timed_out: a count never appeared.
template <u32 MASK>
__device__ __forceinline__ u64 agg3_count_keys(u64 act, u32 k2_base, u32 k1_base, u32 k0_base, u32 cnt_base, u32 &h, u64 w2, u64 w1, u64 w0, u32 &timed_out)
{
    u64 save, t, cur, v1, v0;
    u32 ka2, ka1, ka0, ca, cv, p, spin, tmo = 0;
    const u64 empty = AG_EMPTY;
    const u32 one = 1u;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_movk_i32 %[p], 1\n\t"
        "s_mov_b64 exec, %[act]\n"
        "0:\n\t"
        "v_lshl_add_u32 %[ka2], %[h], 3, %[k2b]\n\t"
        "ds_read_b64 %[cur], %[ka2]\n\t"
        "v_lshl_add_u32 %[ka1], %[h], 3, %[k1b]\n\t"
        "v_lshl_add_u32 %[ka0], %[h], 3, %[k0b]\n\t"
        "v_lshl_add_u32 %[ca], %[h], 2, %[cb]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, -1, %[cur]\n\t"
        "s_cbranch_vccz 1f\n\t"                               // no empty slot among the lanes
        "s_mov_b64 exec, vcc\n\t"
        "ds_cmpst_rtn_b64 %[cur], %[ka2], %[emp], %[w2]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, -1, %[cur]\n\t"                    // claimed: words 1 and 0, then the count that publishes them
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b64 %[ka1], %[w1]\n\t"
        "ds_write_b64 %[ka0], %[w0]\n\t"
        "ds_add_u32 %[ca], %[one]\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"
        "s_cbranch_scc0 2f\n\t"
        "s_mov_b64 exec, %[act]\n"
        "1:\n\t"
        "v_cmp_eq_u64 vcc, %[cur], %[w2]\n\t"                 // word 2 is there: wait for the count, compare words 1 and 0
        "s_cbranch_vccz 4f\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "s_movk_i32 %[spin], 0x7fff\n"
        "5:\n\t"
        "ds_read_b32 %[cv], %[ca]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32 vcc, 0, %[cv]\n\t"                      // not published yet
        "s_cbranch_vccz 6f\n\t"
        "s_sub_u32 %[spin], %[spin], 1\n\t"
        "s_cmp_lg_u32 %[spin], 0\n\t"
        "s_cbranch_scc1 5b\n\t"
        "s_mov_b32 %[tmo], 1\n"
        "6:\n\t"
        "ds_read_b64 %[v1], %[ka1]\n\t"
        "ds_read_b64 %[v0], %[ka0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, %[v1], %[w1]\n\t"
        "v_cmp_eq_u64 %[t], %[v0], %[w0]\n\t"
        "s_and_b64 vcc, vcc, %[t]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_add_u32 %[ca], %[one]\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"
        "s_cbranch_scc0 2f\n\t"
        "s_mov_b64 exec, %[act]\n"
        "4:\n\t"
        "v_add_u32 %[h], %[p], %[h]\n\t"                    // (triangular steps, as agg_count_keys)
        "v_and_b32 %[h], %[mask], %[h]\n\t"
        "s_add_u32 %[p], %[p], 1\n\t"
        "s_cmp_lg_u32 %[p], %[maxp]\n\t"
        "s_cbranch_scc1 0b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[save]"
        : [act] "+s"(act), [h] "+v"(h), [tmo] "+s"(tmo), [save] "=&s"(save), [t] "=&s"(t), [p] "=&s"(p), [spin] "=&s"(spin),
          [cur] "=&v"(cur), [v1] "=&v"(v1), [v0] "=&v"(v0), [cv] "=&v"(cv), [ka2] "=&v"(ka2), [ka1] "=&v"(ka1), [ka0] "=&v"(ka0), [ca] "=&v"(ca)
        : [k2b] "s"(k2_base), [k1b] "s"(k1_base), [k0b] "s"(k0_base), [cb] "s"(cnt_base), [w2] "v"(w2), [w1] "v"(w1), [w0] "v"(w0),
          [emp] "v"(empty), [one] "v"(one), [mask] "n"(MASK), [maxp] "n"(AG_MAX_PROBE + 1)
        : "vcc", "scc", "memory");
    timed_out |= tmo;
    return act;
}

__device__ __forceinline__ bool key3_less(u64 a2, u64 a1, u64 a0, u64 b2, u64 b1, u64 b0)
{
    return a2 < b2 || (a2 == b2 && (a1 < b1 || (a1 == b1 && a0 < b0)));
}

template <int LOG2CAP>
__global__ __launch_bounds__(AG_THREADS) void agg3_finish_kernel(AggArgs a)
{
    constexpr int CAP = 1 << LOG2CAP;
    constexpr int PER = CAP / AG_THREADS;
    __shared__ u64 s_k2[CAP];
    __shared__ u64 s_k1[CAP];
    __shared__ u64 s_k0[CAP];
    __shared__ u32 s_cnt[CAP];
    __shared__ u32 s_bkt[CAP / 2];
    __shared__ u32 s_scr[8];
    __shared__ u32 s_ovf;
    const AggTask &t = a.t[blockIdx.y];
    if (!t.active) return;
    u32 b;
    if (!agg_pick_bin(t, a.nbins, b)) return;
    const int tid = threadIdx.x;
    const u64 s = t.bounds[b], e = t.bounds[b + 1];
    if (e == s) { if (tid == 0) t.bin_cnt[b] = 0; return; }
#pragma unroll
    for (int j = 0; j < PER; ++j) { s_k2[j * AG_THREADS + tid] = AG_EMPTY; s_cnt[j * AG_THREADS + tid] = 0; }
    if (tid == 0) s_ovf = 0;
    __syncthreads();

    constexpr int UNR = 4;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const u32 k2_lds = (u32)(uintptr_t)(LdsPtr)s_k2, k1_lds = (u32)(uintptr_t)(LdsPtr)s_k1, k0_lds = (u32)(uintptr_t)(LdsPtr)s_k0, cnt_lds = (u32)(uintptr_t)(LdsPtr)s_cnt;
    for (u64 i = s + tid; i < e; i += (u64)AG_THREADS * UNR) {
        u64 k0[UNR], k1[UNR], k2[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const u64 idx = i + (u64)u * AG_THREADS;
            const bool ok = idx < e;
            k0[u] = ok ? t.keys[idx * 3] : 0; k1[u] = ok ? t.keys[idx * 3 + 1] : 0; k2[u] = ok ? t.keys[idx * 3 + 2] : AG_EMPTY;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const u64 w0 = k0[u], w1 = k1[u], w2 = k2[u];
            const u64 act = __ballot(w2 != AG_EMPTY);
            if (act == 0) continue;                       // (uniform)
            const u64 m = w0 ^ (w1 >> 7) ^ (w1 << 23) ^ (w2 >> 9) ^ (w2 << 21);
            const u32 x = (u32)(m >> 32) ^ (u32)m;
            u32 tmo = 0, h = (x * 0x9E3779B1u) >> (32 - LOG2CAP);
            if (agg3_count_keys<(u32)CAP - 1u>(act, k2_lds, k1_lds, k0_lds, cnt_lds, h, w2, w1, w0, tmo) != 0 || tmo) s_ovf = 1;      // (uniform)
        }
        if (__hip_atomic_load(&s_ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
    }
    __syncthreads();
    if (s_ovf) {
        if (tid == 0) agg_bin_overflow(t, a.nbins, b, (u32)CAP);
        return;
    }

    // ---- compact, order by (word 2, word 1, word 0) ------------------------------------------------------------------
    u32 D;
    {
        u64 m2[PER], m1[PER], m0[PER]; u32 mc[PER];
        u32 occ = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int sl = tid * PER + j;
            m2[j] = s_k2[sl]; m1[j] = s_k1[sl]; m0[j] = s_k0[sl]; mc[j] = s_cnt[sl]; occ += m2[j] != AG_EMPTY;
        }
        u32 o = block_excl_scan_256<u32>(occ, s_scr, &D);
#pragma unroll
        for (int j = 0; j < PER; ++j) if (m2[j] != AG_EMPTY) { s_k2[o] = m2[j]; s_k1[o] = m1[j]; s_k0[o] = m0[j]; s_cnt[o] = mc[j]; ++o; }
    }
    __syncthreads();
    D = (u32)__builtin_amdgcn_readfirstlane((int)D);
    if (D <= (u32)AG_THREADS) {
        u64 k2 = 0, k1 = 0, k0 = 0; u32 c = 0, r = 0;
        if ((u32)tid < D) {
            k2 = s_k2[tid]; k1 = s_k1[tid]; k0 = s_k0[tid]; c = s_cnt[tid];
            for (u32 j = 0; j < D; ++j) r += key3_less(s_k2[j], s_k1[j], s_k0[j], k2, k1, k0);
        }
        __syncthreads();
        if ((u32)tid < D) { s_k2[r] = k2; s_k1[r] = k1; s_k0[r] = k0; s_cnt[r] = c; }
        __syncthreads();
    } else {
        u64 *const kw[3] = {s_k0, s_k1, s_k2};
        agg_order_many<LOG2CAP, 3, false>(D, kw, s_cnt, nullptr, s_bkt, s_scr, a.top_sig);
    }

    // ---- filter, entries {word 0, word 1, word 2, count} in key order to the bin's slots -----------------------------
    u32 kept = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const u32 i = tid * PER + j;
        if (i < D) { const u32 c = s_cnt[i]; kept += (c >= a.lower && c <= a.upper); }
    }
    u32 tot;
    const u32 w = block_excl_scan_256<u32>(kept, s_scr, &tot);
    u64 *dst = t.scratch + ((s >> t.slot_shift) + w) * 4;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const u32 i = tid * PER + j;
        if (i < D) {
            const u32 c = s_cnt[i];
            if (c >= a.lower && c <= a.upper) { dst[0] = s_k0[i]; dst[1] = s_k1[i]; dst[2] = s_k2[i]; dst[3] = (u64)c; dst += 4; }
        }
    }
    if (tid == 0) t.bin_cnt[b] = tot;
}

// ------------------------------------------------------------------------------------------------------------
// EXTENSION (every k-mer carries (PosInRead, ReadId)): the same 16-bit prefix bins, but the records must come out
// GROUPED by key, because an entry owns a slice of the task's payload array (count_sorted_kmers copies the run's
// pos/rid, reference src/kmerops.cpp:1430-1437).  Two sweeps over the bin instead of ordering its records:
//   1. count the distinct keys in the LDS table (as above), order them, prefix-sum the counts in key order:
//      group g of the bin starts at record offset goff[g] of the bin's range;
//   2. every record's place is goff[group] + (records of its key counted before it).  Bins of up to 4096 records -- all
//      but the outliers -- keep {slot, earlier records} of every record in registers from the first sweep (the returning
//      add of the count), so the second sweep is one pass over the payloads: they go to their places in an LDS stage,
//      window by window, and leave as contiguous pos / rid stores (scattered 4-byte stores straight to the arrays cost
//      74 of this kernel's 117 ms per benchmark step).  Larger bins read their records again, find the group through
//      the table, take the next free place of the group with an LDS atomic and store directly.  The order of the
//      payloads inside one k-mer is free (the reference's sorts are not stable either).
// Entries {key words, count} and their payload offsets go to per-bin slots and are compacted afterwards.
// ------------------------------------------------------------------------------------------------------------
struct AggExtTask {
    const u64 *keys, *vals; u64 n;
    u64 *bounds, *bin_cnt;
    u64 *scratch_e;                // {key, count} of kept entries, bin b from entry slot (bounds[b] >> slot_shift)
    u64 *scratch_p;                // their payload offsets, same slots
    u32 slot_shift, active;
    u32 *pos; int32_t *rid;        // [n] payloads grouped by key, bins in order
    u64 payoff_add;                // offset of this task's payload range in the rank's payload arrays
    u32 *flags;
    // the table ladder bin by bin, as in AggTask: bins that overflow this launch's table are listed for the next one
    const u32 *bin_list; const u32 *bin_list_n;
    u32 *ovf_list; u32 *ovf_n;
};
struct AggExtArgs { AggExtTask t[AG_BATCH]; u32 lower, upper; u32 nbins; int shift; int nw; int top_bits; int top_sig; };     // nw, top_bits: as in AggArgs (nw = 0 reads as 1)

__global__ __launch_bounds__(AG_THREADS) void bin_bounds_ext_kernel(AggExtArgs a)
{
    const AggExtTask &t = a.t[blockIdx.y];
    const u32 b = blockIdx.x * AG_THREADS + threadIdx.x;
    if (!t.active || b > a.nbins) return;
    u64 lo = 0, hi = t.n;
    if (b == a.nbins) lo = t.n;
    else while (lo < hi) {
        const u64 mid = lo + ((hi - lo) >> 1);
        const int nw = a.nw ? a.nw : 1;
        const u64 *rec = t.keys + mid * nw;
        const u32 bin = (a.top_bits > 0 && a.top_bits < 16) ? (((u32)(rec[nw - 1] >> (64 - a.top_bits)) << (16 - a.top_bits)) | (u32)(rec[nw - 2] >> (48 + a.top_bits)))
                                                             : (u32)(rec[nw - 1] >> a.shift);
        if (bin < b) lo = mid + 1; else hi = mid;
    }
    t.bounds[b] = lo;
}

// The kernel, for keys of NW = 1 .. 3 words (records {word 0 .. word NW-1}, entries {words, count}).  14 + 16 NW bytes of LDS per
// slot -- the table's keys, the distinct keys compacted / ordered with their counts and origin slots, slot -> first record of the
// slot's group, records per slot --, all of which becomes the payload stage once every record of a bin in registers knows its
// place.  Tables: 1024 slots first; 4096 (one word) or 2048 (two, three words) for the bins the first could not hold.  Slots are
// claimed with agg_count_keys / agg2_count_keys / agg3_count_keys.  (Multi-word keys with EXTENSION took 13 - 20 full LSD passes over
// records of 24 - 32 bytes until round 2.)
template <int NW> __device__ __forceinline__ bool keyw_less(const u64 (&x)[NW], const u64 (&y)[NW])
{
#pragma unroll
    for (int w = NW - 1; w > 0; --w) if (x[w] != y[w]) return x[w] < y[w];
    return x[0] < y[0];
}
template <int LOG2CAP, int NW>
__global__ __launch_bounds__(AG_THREADS) void agg_ext_kernel(AggExtArgs a)
{
    static_assert(NW >= 1 && NW <= 3, "keys of one to three words");
    constexpr int CAP = 1 << LOG2CAP;
    constexpr int PER = CAP / AG_THREADS;
    // records per lane whose slots stay in registers: bins of up to 8192 records (the bins of prefixes that start with A hold
    // twice the average: canonical k-mers), loaded in batches of UNR
    constexpr int UNR = NW == 1 ? 16 : 8, REGS = 32, NBAT = REGS / UNR;
    constexpr u32 STAGE = (u32)CAP * (14u + 16u * NW) / 8u;
    __shared__ __attribute__((aligned(16))) u64 s_raw[STAGE];
    __shared__ u32 s_bkt[CAP / 2];
    __shared__ u32 s_scr[8];
    __shared__ u32 s_ovf;
    // raw: [ordered keys NW x CAP x 8][counts CAP x 4][origin slots CAP x 2][table keys NW x CAP x 8][group offsets CAP x 4][table counts CAP x 4]
    u64 *s_k = s_raw;                                                   // word w of entry i: s_k[w * CAP + i]
    u32 *s_cnt = reinterpret_cast<u32 *>(s_raw + (size_t)NW * CAP);
    u16 *s_slot = reinterpret_cast<u16 *>(s_cnt + CAP);
    u64 *s_tk = reinterpret_cast<u64 *>(s_slot + CAP);                  // (CAP x 6 bytes behind 8-byte aligned data: CAP is a multiple of 4)
    u32 *s_soff = reinterpret_cast<u32 *>(s_tk + (size_t)NW * CAP);
    u32 *s_tcnt = s_soff + CAP;
    const AggExtTask &t = a.t[blockIdx.y];
    if (!t.active) return;
    u32 b = blockIdx.x;
    if (t.bin_list) { if (b >= *t.bin_list_n || b >= a.nbins) return; b = t.bin_list[b]; if (b >= a.nbins) return; }
    const int tid = threadIdx.x;
    const u64 s = t.bounds[b], e = t.bounds[b + 1];
    if (e == s) { if (tid == 0) t.bin_cnt[b] = 0; return; }
    const bool in_regs = e - s <= (u64)AG_THREADS * REGS;              // (uniform)
    const u32 nrec = in_regs ? (u32)(e - s) : 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int sl = j * AG_THREADS + tid;
        s_tk[(NW - 1) * CAP + sl] = AG_EMPTY; s_tcnt[sl] = 0;
        if (NW == 2) s_tk[sl] = AG_EMPTY;                               // (two words: word 0 doubles as the publication flag, agg2_count_keys)
    }
    if (tid == 0) s_ovf = 0;
    __syncthreads();

    // ---- 1. first sweep: count ----------------------------------------------------------------------------------
    typedef __attribute__((address_space(3))) void *LdsPtr;
    const u32 tk_lds = (u32)(uintptr_t)(LdsPtr)s_tk, cnt_lds = (u32)(uintptr_t)(LdsPtr)s_tcnt;
    u32 where[REGS];
#pragma unroll
    for (int u = 0; u < REGS; ++u) where[u] = 0;
    auto slot_of = [&](const u64 (&k)[NW]) -> u32 {
        if (NW == 1) return agg_slot<LOG2CAP>(k[0]);
        const u64 m = NW == 2 ? (k[0] ^ (k[NW - 1] >> 9) ^ (k[NW - 1] << 21)) : (k[0] ^ (k[NW > 1 ? 1 : 0] >> 7) ^ (k[NW > 1 ? 1 : 0] << 23) ^ (k[NW - 1] >> 9) ^ (k[NW - 1] << 21));
        const u32 x = (u32)(m >> 32) ^ (u32)m;
        return (x * 0x9E3779B1u) >> (32 - LOG2CAP);
    };
    auto count_batch = [&](u64 i, u32 *wh) {
        u64 k[UNR][NW];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const u64 idx = i + (u64)u * AG_THREADS;
#pragma unroll
            for (int w = 0; w < NW; ++w) k[u][w] = idx < e ? t.keys[idx * NW + w] : (w == NW - 1 ? AG_EMPTY : 0);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const u64 act = __ballot(k[u][NW - 1] != AG_EMPTY);
            if (act == 0) continue;                       // (uniform)
            u32 h = slot_of(k[u]), tmo = 0;
            u64 left;
            if (NW == 1) left = agg_count_keys<(u32)CAP - 1u>(act, tk_lds, cnt_lds, h, k[u][0]);
            else if (NW == 2) left = agg2_count_keys<(u32)CAP - 1u>(act, tk_lds + (u32)CAP * 8u, tk_lds, cnt_lds, h, k[u][NW - 1], k[u][0], tmo);
            else left = agg3_count_keys<(u32)CAP - 1u>(act, tk_lds + 2u * (u32)CAP * 8u, tk_lds + (u32)CAP * 8u, tk_lds, cnt_lds, h, k[u][NW - 1], k[u][NW > 1 ? 1 : 0], k[u][0], tmo);
            if (left != 0 || tmo) s_ovf = 1;
            if (wh) wh[u] = h;
        }
    };
    if (in_regs) {
#pragma unroll
        for (int bt_ = 0; bt_ < NBAT; ++bt_)
            if (s + (u64)bt_ * AG_THREADS * UNR < e) count_batch(s + tid + (u64)bt_ * AG_THREADS * UNR, where + bt_ * UNR);      // (uniform)
    } else {
        for (u64 i = s + tid; i < e; i += (u64)AG_THREADS * UNR) {
            count_batch(i, nullptr);
            if (__hip_atomic_load(&s_ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
    }
    __syncthreads();
    if (s_ovf) {
        if (tid == 0) {
            if (t.ovf_list) { const u32 at = atomicAdd(t.ovf_n, 1u); if (at < a.nbins) t.ovf_list[at] = b; else atomicOr(t.flags, (u32)AG_FLAG_OVERFLOW); }
            else atomicOr(t.flags, (u32)AG_FLAG_OVERFLOW);
            t.bin_cnt[b] = 0;
        }
        return;
    }

    // ---- 2. distinct keys in key order, group offsets --------------------------------------------------------------
    u32 D;
    {
        u32 occ = 0;
        u64 mk[PER][NW]; u32 mc[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int sl = tid * PER + j;
#pragma unroll
            for (int w = 0; w < NW; ++w) mk[j][w] = s_tk[w * CAP + sl];
            mc[j] = s_tcnt[sl]; occ += mk[j][NW - 1] != AG_EMPTY;
        }
        u32 o = block_excl_scan_256<u32>(occ, s_scr, &D);
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (mk[j][NW - 1] != AG_EMPTY) {
#pragma unroll
                for (int w = 0; w < NW; ++w) s_k[w * CAP + o] = mk[j][w];
                s_cnt[o] = mc[j]; s_slot[o] = (u16)(tid * PER + j); ++o;
            }
    }
    __syncthreads();
    D = (u32)__builtin_amdgcn_readfirstlane((int)D);
    if (D <= (u32)AG_THREADS) {
        u64 k[NW]; u32 c = 0, r = 0; u16 sl = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = 0;
        if ((u32)tid < D) {
#pragma unroll
            for (int w = 0; w < NW; ++w) k[w] = s_k[w * CAP + tid];
            c = s_cnt[tid]; sl = s_slot[tid];
            for (u32 j = 0; j < D; ++j) {
                u64 o[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) o[w] = s_k[w * CAP + j];
                r += keyw_less<NW>(o, k);
            }
        }
        __syncthreads();
        if ((u32)tid < D) {
#pragma unroll
            for (int w = 0; w < NW; ++w) s_k[w * CAP + r] = k[w];
            s_cnt[r] = c; s_slot[r] = sl;
        }
        __syncthreads();
    } else {
        u64 *kw[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) kw[w] = s_k + (size_t)w * CAP;
        u64 *const (&kwc)[NW] = reinterpret_cast<u64 *const (&)[NW]>(kw);
        agg_order_many<LOG2CAP, NW, true>(D, kwc, s_cnt, s_slot, s_bkt, s_scr, a.top_sig);
    }
    u32 kept = 0, gof[PER];
    {
        u32 csum = 0, cv[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const u32 i = tid * PER + j;
            cv[j] = i < D ? s_cnt[i] : 0;
            csum += cv[j];
            kept += (i < D && cv[j] >= a.lower && cv[j] <= a.upper);
        }
        u32 go = block_excl_scan_256<u32>(csum, s_scr, nullptr);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const u32 i = tid * PER + j;
            gof[j] = go;
            if (i < D) s_soff[s_slot[i]] = go;
            go += cv[j];
        }
    }
    u32 tot;
    const u32 w_ = block_excl_scan_256<u32>(kept, s_scr, &tot);
    {
        const u64 slot0 = (s >> t.slot_shift) + w_;
        u64 *de = t.scratch_e + slot0 * (NW + 1); u64 *dp = t.scratch_p + slot0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const u32 i = tid * PER + j;
            if (i < D) {
                const u32 c = s_cnt[i];
                if (c >= a.lower && c <= a.upper) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) de[w] = s_k[w * CAP + i];
                    de[NW] = (u64)c; dp[0] = t.payoff_add + s + gof[j]; de += NW + 1; ++dp;
                }
            }
        }
    }
    if (tid == 0) t.bin_cnt[b] = tot;
    __syncthreads();

    // ---- 3. second sweep: every payload to its place ---------------------------------------------------------------
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < REGS; ++u) {
            const u32 r = (u32)tid + (u32)u * AG_THREADS;
            where[u] = r < nrec ? s_soff[where[u]] + atomicSub(&s_tcnt[where[u]], 1u) - 1u : ~0u;
        }
        __syncthreads();
        for (u32 w0 = 0; w0 < nrec; w0 += STAGE) {
#pragma unroll
            for (int bt_ = 0; bt_ < NBAT; ++bt_) {
                if ((u32)bt_ * AG_THREADS * UNR >= nrec) continue;                                  // (uniform)
                u64 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) { const u32 r = (u32)tid + (u32)(bt_ * UNR + u) * AG_THREADS; v[u] = t.vals[s + (r < nrec ? r : 0u)]; }
#pragma unroll
                for (int u = 0; u < UNR; ++u) { const u32 o = where[bt_ * UNR + u]; if (o - w0 < STAGE) s_raw[o - w0] = v[u]; }
            }
            __syncthreads();
            const u32 n = nrec - w0 < STAGE ? nrec - w0 : STAGE;
            for (u32 i = tid; i < n; i += AG_THREADS) {
                const u64 x = s_raw[i];
                t.pos[s + w0 + i] = (u32)x; t.rid[s + w0 + i] = (int32_t)(x >> 32);
            }
            __syncthreads();
        }
    } else {
        for (u64 i = s + tid; i < e; i += AG_THREADS) {                 // (a bin too large for registers: every record looks its slot up again)
            u64 k[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) k[w] = t.keys[i * NW + w];
            const u64 v = t.vals[i];
            u32 h = slot_of(k);
            for (;;) {
                bool same = true;
#pragma unroll
                for (int w = 0; w < NW; ++w) same = same && s_tk[w * CAP + h] == k[w];
                if (same) break;
                h = (h + 1) & (CAP - 1);
            }
            const u64 o = s + s_soff[h] + atomicSub(&s_tcnt[h], 1u) - 1u;
            t.pos[o] = (u32)v; t.rid[o] = (int32_t)(v >> 32);
        }
    }
}

// entries and payload offsets from the per-bin slots to their final places; count histogram.  One wave per bin.
struct AggExtCompactArgs { const u64 *scratch_e[AG_BATCH]; const u64 *scratch_p[AG_BATCH]; const u64 *bounds[AG_BATCH]; const u64 *bin_off[AG_BATCH];
                           u64 *entries[AG_BATCH]; u64 *payoff[AG_BATCH]; u32 slot_shift; u64 *histo; u32 histo_len; u32 nbins; u32 ew; };     // ew: words per entry (0 reads as 2)
__global__ __launch_bounds__(AG_THREADS) void agg_ext_compact_kernel(AggExtCompactArgs ca)
{
    __shared__ u32 s_hist[AG_LDS_HIST];
    u64 *entries = ca.entries[blockIdx.y];
    if (!entries) return;
    const u64 *se = ca.scratch_e[blockIdx.y], *sp = ca.scratch_p[blockIdx.y], *bounds = ca.bounds[blockIdx.y], *bin_off = ca.bin_off[blockIdx.y];
    u64 *payoff = ca.payoff[blockIdx.y];
    for (int i = threadIdx.x; i < AG_LDS_HIST; i += AG_THREADS) s_hist[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (u32 b = blockIdx.x * 4 + wave; b < ca.nbins; b += gridDim.x * 4) {
        const u64 o = bin_off[b];
        const u64 cnt = bin_off[b + 1] - o;
        const u64 slot0 = bounds[b] >> ca.slot_shift;
        const u64 ew = ca.ew ? ca.ew : 2;
        for (u64 i = lane; i < cnt * ew; i += 64) {
            const u64 v = se[slot0 * ew + i];
            entries[o * ew + i] = v;
            if (i % ew == ew - 1) {
                if (v < (u64)AG_LDS_HIST) atomicAdd(&s_hist[(u32)v], 1u);
                else if (v < ca.histo_len) atomicAdd((unsigned long long *)&ca.histo[v], 1ULL);
            }
        }
        for (u64 i = lane; i < cnt; i += 64) payoff[o + i] = sp[slot0 + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < AG_LDS_HIST; i += AG_THREADS) {
        const u32 c = s_hist[i];
        if (c && (u32)i < ca.histo_len) atomicAdd((unsigned long long *)&ca.histo[i], (unsigned long long)c);
    }
}

// ------------------------------------------------------------------------------------------------------------
// One scatter pass, then aggregate: bins of the top 8 key bits.  With tasks of ~2^24 k-mers a bin holds ~65 000
// records and ~2 500 distinct keys (32 x coverage): too many records for LDS, but the hash table only has to hold
// the DISTINCT keys.  One workgroup of 1024 threads per bin, 8192 slots (96 KB of the CU's 160 KB LDS): the bin
// is streamed once from HBM, the distinct keys are compacted and ordered with a bitonic network in LDS, filtered
// and written in key order.  The per-bin fixed work (table init, compaction, ordering, output) is spread over
// ~65 000 records instead of ~3 000, and the second scatter pass (16 B of HBM traffic per k-mer) is not needed.
// More distinct keys than the table takes (low coverage, or a skewed bin) raise AG_FLAG_OVERFLOW: the host sorts
// that task on the next 8 bits as well and finishes it with agg_finish_kernel.
// ------------------------------------------------------------------------------------------------------------
constexpr int AGB_THREADS = 1024;
constexpr int AGB_LOG2CAP = 13;
constexpr int AGB_CAP = 1 << AGB_LOG2CAP;
constexpr int AGB_MAX_LOAD = AGB_CAP * 3 / 4;      // distinct keys accepted (beyond that probes get long: overflow)
constexpr int AGB_NBKT = 4096;                     // buckets of the in-LDS counting sort: the 12 key bits below the 8-bit bin prefix

template <typename T>
__device__ __forceinline__ T block_excl_scan_1024(T v, T *scratch /* >= 16 */, T *total)
{
    const int lane = lane_id(), w = threadIdx.x >> 6;
    T inc = wave_incl_scan(v);
    if (lane == WAVE - 1) scratch[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { T x = scratch[i]; if (i < w) base += x; tot += x; }
    __syncthreads();
    if (total) *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(AGB_THREADS) void agg_big_kernel(AggArgs a)
{
    constexpr int CAP = AGB_CAP;
    constexpr int PER = CAP / AGB_THREADS;          // 8 slots per thread
    __shared__ u64 s_key[CAP];
    __shared__ u32 s_cnt[CAP];
    __shared__ u32 s_hist[AGB_NBKT];
    __shared__ u32 s_scr[16];
    __shared__ u32 s_ovf;
    const AggTask &t = a.t[blockIdx.y];
    if (!t.active) return;
    u32 b;
    if (!agg_pick_bin(t, a.nbins, b)) return;
    const int tid = threadIdx.x;
    const int bkt_shift = a.shift - 12;                 // the 12 key bits right below the bin prefix
    const u64 s = t.bounds[b], e = t.bounds[b + 1];
    if (e == s) { if (tid == 0) t.bin_cnt[b] = 0; return; }
#pragma unroll
    for (int j = 0; j < PER; ++j) { s_key[j * AGB_THREADS + tid] = AG_EMPTY; s_cnt[j * AGB_THREADS + tid] = 0; }
    if (tid == 0) s_ovf = 0;
    __syncthreads();

    // ---- 1. stream the bin: 8 loads per lane in flight ---------------------------------------------------------
    constexpr int UNR = 8;
    for (u64 i = s + tid; i < e; i += (u64)AGB_THREADS * UNR) {
        u64 k[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { const u64 idx = i + (u64)u * AGB_THREADS; k[u] = idx < e ? t.keys[idx] : AG_EMPTY; }
        // first probes of all 8 records together (independent LDS reads: one latency instead of eight); a record whose key
        // already sits in its home slot -- nearly all of them after a key's first occurrence -- is one more LDS atomic
        u32 hh[UNR]; u64 cur0[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { const u32 x = (u32)(k[u] >> 32) ^ (u32)k[u]; hh[u] = (x * 0x9E3779B1u) >> (32 - AGB_LOG2CAP); }
#pragma unroll
        for (int u = 0; u < UNR; ++u) cur0[u] = __hip_atomic_load(&s_key[hh[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (k[u] == AG_EMPTY) continue;
            if (cur0[u] == k[u]) { atomicAdd(&s_cnt[hh[u]], 1u); continue; }
            u32 h = hh[u];
            bool done = false;
            for (int p = 0; p < AG_MAX_PROBE; ++p) {
                u64 cur = __hip_atomic_load(&s_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == AG_EMPTY) cur = atomicCAS((unsigned long long *)&s_key[h], (unsigned long long)AG_EMPTY, (unsigned long long)k[u]);
                if (cur == AG_EMPTY || cur == k[u]) { atomicAdd(&s_cnt[h], 1u); done = true; break; }
                h = (h + 1) & (CAP - 1);
            }
            if (!done) s_ovf = 1;
        }
    }
    __syncthreads();

    // ---- 2. compact the occupied slots ---------------------------------------------------------------------------
    u32 D;
    {
        u64 mk[PER]; u32 mc[PER];
        u32 occ = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { mk[j] = s_key[tid * PER + j]; mc[j] = s_cnt[tid * PER + j]; occ += mk[j] != AG_EMPTY; }
        u32 o = block_excl_scan_1024<u32>(occ, s_scr, &D);
#pragma unroll
        for (int j = 0; j < PER; ++j) if (mk[j] != AG_EMPTY) { s_key[o] = mk[j]; s_cnt[o] = mc[j]; ++o; }
    }
    __syncthreads();
    if (s_ovf || D > (u32)AGB_MAX_LOAD) {
        if (tid == 0) agg_bin_overflow(t, a.nbins, b, (u32)AGB_CAP);
        return;
    }

    // ---- 3. order the distinct keys ------------------------------------------------------------------------------
    // Up to CAP / 2 keys (the usual case): counting sort on the 12 bits below the bin prefix into the idle upper half of
    // the table (the keys of a bin are close to uniform there: about one key per bucket), then every key ranks itself
    // inside its bucket.  ~10 LDS operations per key instead of the ~400 of a sorting network over {key, count} pairs;
    // a bucket that collects many keys (a shared 10-base prefix) only costs its own square.
    if (D <= (u32)(CAP / 2)) {
        constexpr int EPT = CAP / 2 / AGB_THREADS;          // 4 elements per thread
        for (int i = tid; i < AGB_NBKT; i += AGB_THREADS) s_hist[i] = 0;
        __syncthreads();
        u64 ek[EPT]; u32 ec[EPT], er[EPT];
#pragma unroll
        for (int x = 0; x < EPT; ++x) {
            const u32 i = x * AGB_THREADS + tid;
            ek[x] = AG_EMPTY; ec[x] = 0; er[x] = 0;
            if (i < D) { ek[x] = s_key[i]; ec[x] = s_cnt[i]; er[x] = atomicAdd(&s_hist[(u32)(ek[x] >> bkt_shift) & (AGB_NBKT - 1)], 1u); }
        }
        __syncthreads();
        {
            u32 v[AGB_NBKT / AGB_THREADS], sum = 0;
#pragma unroll
            for (int x = 0; x < AGB_NBKT / AGB_THREADS; ++x) { v[x] = s_hist[tid * (AGB_NBKT / AGB_THREADS) + x]; sum += v[x]; }
            u32 ex = block_excl_scan_1024<u32>(sum, s_scr, nullptr);
#pragma unroll
            for (int x = 0; x < AGB_NBKT / AGB_THREADS; ++x) { s_hist[tid * (AGB_NBKT / AGB_THREADS) + x] = ex; ex += v[x]; }
        }
        __syncthreads();
        u64 *ok = s_key + CAP / 2; u32 *oc = s_cnt + CAP / 2;
#pragma unroll
        for (int x = 0; x < EPT; ++x)
            if (ek[x] != AG_EMPTY) { const u32 p = s_hist[(u32)(ek[x] >> bkt_shift) & (AGB_NBKT - 1)] + er[x]; ok[p] = ek[x]; oc[p] = ec[x]; }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < EPT; ++x) {
            if (ek[x] == AG_EMPTY) continue;
            const u32 bk = (u32)(ek[x] >> bkt_shift) & (AGB_NBKT - 1);
            const u32 b0 = s_hist[bk], b1 = (bk + 1 < (u32)AGB_NBKT) ? s_hist[bk + 1] : D;
            u32 r = b0;
            for (u32 q = b0; q < b1; ++q) r += ok[q] < ek[x];
            s_key[r] = ek[x]; s_cnt[r] = ec[x];
        }
        __syncthreads();
    } else {
    u32 P = 2; while (P < D) P <<= 1;
    for (u32 i = D + tid; i < P; i += AGB_THREADS) { s_key[i] = AG_EMPTY; s_cnt[i] = 0; }
    __syncthreads();
    for (u32 kk = 2; kk <= P; kk <<= 1) {
        for (u32 j = kk >> 1; j > 0; j >>= 1) {
            // each thread owns the compare-exchanges whose lower index has bit j clear: i = insert a 0 at bit log2(j)
            for (u32 x = tid; x < (P >> 1); x += AGB_THREADS) {
                const u32 i = ((x & ~(j - 1)) << 1) | (x & (j - 1));
                const u32 q = i | j;
                const u64 ki = s_key[i], kq = s_key[q];
                const bool up = (i & kk) == 0;
                if ((ki > kq) == up) { const u32 ci = s_cnt[i], cq = s_cnt[q]; s_key[i] = kq; s_key[q] = ki; s_cnt[i] = cq; s_cnt[q] = ci; }
            }
            __syncthreads();
        }
    }
    }

    // ---- 4. filter, entries in key order to the bin's slots ------------------------------------------------------
    u32 kept = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const u32 i = tid * PER + j;
        if (i < D) { const u32 c = s_cnt[i]; kept += (c >= a.lower && c <= a.upper); }
    }
    u32 tot;
    const u32 w = block_excl_scan_1024<u32>(kept, s_scr, &tot);
    u64 *dst = t.scratch + ((s >> t.slot_shift) + w) * 2;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const u32 i = tid * PER + j;
        if (i < D) {
            const u32 c = s_cnt[i];
            if (c >= a.lower && c <= a.upper) { dst[0] = s_key[i]; dst[1] = (u64)c; dst += 2; }
        }
    }
    if (tid == 0) t.bin_cnt[b] = tot;
}

// exclusive scan of bin_cnt[AG_BINS] of every active task (total behind the last bin); one workgroup per task
__global__ __launch_bounds__(AG_THREADS) void agg_scan_kernel(AggArgs a)
{
    __shared__ u64 s_scr[8];
    __shared__ u64 s_carry;
    const AggTask &t = a.t[blockIdx.x];
    if (!t.active) return;
    constexpr int IPT = 8;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 b = 0; b < a.nbins; b += AG_THREADS * IPT) {
        const u32 t0 = b + threadIdx.x * IPT;
        u64 v[IPT], sum = 0;
#pragma unroll
        for (int i = 0; i < IPT; ++i) { v[i] = (t0 + i < a.nbins) ? t.bin_cnt[t0 + i] : 0; sum += v[i]; }
        u64 tot;
        u64 ex = block_excl_scan_256<u64>(sum, s_scr, &tot) + s_carry;
#pragma unroll
        for (int i = 0; i < IPT; ++i) { if (t0 + i < a.nbins) t.bin_off[t0 + i] = ex; ex += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) t.bin_off[a.nbins] = s_carry;
}

// Moves the kept entries from the per-bin slots to their final place (bin_off = exclusive scan of bin_cnt,
// bin_off[AG_BINS] = total) and builds the count histogram.  One wave per bin, persistent workgroups.
struct AggCompactArgs { const u64 *scratch[AG_BATCH]; const u64 *bounds[AG_BATCH]; const u64 *bin_off[AG_BATCH]; u64 *entries[AG_BATCH]; u32 slot_shift; u64 *histo; u32 histo_len; u32 nbins; u32 ew; };   // ew = words per entry (key words + count)
__global__ __launch_bounds__(AG_THREADS) void agg_compact_kernel(AggCompactArgs ca)
{
    __shared__ u32 s_hist[AG_LDS_HIST];
    u64 *entries = ca.entries[blockIdx.y];
    if (!entries) return;                               // task not handled here (empty, or redone the long way)
    const u64 *scratch = ca.scratch[blockIdx.y], *bounds = ca.bounds[blockIdx.y], *bin_off = ca.bin_off[blockIdx.y];
    const u32 slot_shift = ca.slot_shift, histo_len = ca.histo_len;
    u64 *histo = ca.histo;
    for (int i = threadIdx.x; i < AG_LDS_HIST; i += AG_THREADS) s_hist[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (u32 b = blockIdx.x * 4 + wave; b < ca.nbins; b += gridDim.x * 4) {
        const u64 o = bin_off[b];
        const u64 cnt = bin_off[b + 1] - o;
        const u64 *src = scratch + (bounds[b] >> slot_shift) * ca.ew;
        for (u64 i = lane; i < cnt * ca.ew; i += 64) {
            const u64 v = src[i];
            entries[o * ca.ew + i] = v;
            if (i % ca.ew == ca.ew - 1) {
                if (v < (u64)AG_LDS_HIST) atomicAdd(&s_hist[(u32)v], 1u);
                else if (v < histo_len) atomicAdd((unsigned long long *)&histo[v], 1ULL);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < AG_LDS_HIST; i += AG_THREADS) {
        const u32 c = s_hist[i];
        if (c && (u32)i < histo_len) atomicAdd((unsigned long long *)&histo[i], (unsigned long long)c);
    }
}

} // namespace hsk
