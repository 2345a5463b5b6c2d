/*
 * hsk_oracle.c -- CPU restatement of the HySortK k-mer counting hot path (TEST INFRASTRUCTURE).
 *
 * This file is the parity oracle.  It is NOT on the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (hysortk_amd/csrc, libhsk.so) never links or calls anything in oracle/.
 *
 * Every function restates, in plain C, the algorithm of the reference file:line it cites
 * (paths relative to the reference checkout, CornellHPC/HySortK @ 2025-02-02).  All
 * reference compile-time macros (KMER_SIZE, MINIMIZER_SIZE, LOWER/UPPER_KMER_FREQ,
 * EXTENSION) are runtime arguments here.
 *
 * Parity status: PINNED.  tests/golden/ holds outputs of the real reference built by
 * oracle/build_ref.sh (murmur KATs, canonical k-mers, destinations, supermers, full
 * (k-mer,count) lists, EXT payloads, histogram text); tests/test_oracle_golden.py checks this
 * file against every one of them.
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC -o oracle/libhsk_oracle.so oracle/hsk_oracle.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HSKO_MAXW 3          /* Kmer<1..3>: kmer.hpp:343-345 */
#define HSKO_MAX_SUPERMER_LEN 250 /* supermer.hpp:20 */

typedef struct { uint64_t w[HSKO_MAXW]; } hsko_mer;

/* ------------------------------------------------------------------------------------------
 * a1. 2-bit packing: DnaSeq::compress (src/dnaseq.cpp:9-31), codetab (include/dnaseq.hpp:138)
 * A/a/N/n -> 0, C/c -> 1, G/g -> 2, T/t -> 3, anything else -> 4 (which, shifted as uint8_t,
 * corrupts the neighbouring bits exactly like the reference: dnaseq.cpp:23-25).
 * ---------------------------------------------------------------------------------------- */
static uint8_t hsko_code(char c)
{
    switch (c) {
    case 'A': case 'a': case 'N': case 'n': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

size_t hsko_bytesneeded(size_t n) { return (n + 3) / 4; } /* dnaseq.hpp:126 */

void hsko_pack(const char *s, size_t len, uint8_t *mem)
{
    size_t nbytes = hsko_bytesneeded(len);
    int remain = (int)(4 * nbytes - len);
    for (size_t b = 0; b < nbytes; ++b) {
        uint8_t byte = 0;
        int left = (b != nbytes - 1) ? 4 : 4 - remain;
        for (int i = 0; i < left; ++i) {
            uint8_t code = hsko_code(s[4 * b + i]);
            uint8_t shift = (uint8_t)(code << (6 - 2 * i));
            byte |= shift;
        }
        mem[b] = byte;
    }
}

/* DnaSeq::operator[] (src/dnaseq.cpp:50-56) */
static inline int hsko_base(const uint8_t *mem, size_t i)
{
    return (mem[i / 4] >> (6 - 2 * (i % 4))) & 3;
}

/* ------------------------------------------------------------------------------------------
 * a3. MurmurHash3_x64_128, seed 313, low word (src/hashfuncs.cpp:42-114, :233-238)
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

uint64_t hsko_murmur64(const void *key, uint32_t len)
{
    const uint8_t *data = (const uint8_t *)key;
    const uint32_t nblocks = len / 16;
    uint64_t h1 = 313, h2 = 313;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (uint32_t i = 0; i < nblocks; i++) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    uint32_t rem = len & 15;
    if (rem > 8) {
        for (uint32_t i = rem; i > 8; --i) k2 ^= (uint64_t)tail[i - 1] << (8 * (i - 9));
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (rem > 0) {
        uint32_t top = rem > 8 ? 8 : rem;
        for (uint32_t i = top; i > 0; --i) k1 ^= (uint64_t)tail[i - 1] << (8 * (i - 1));
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    return h1;
}

/* ------------------------------------------------------------------------------------------
 * a2. Kmer<N> (include/kmer.hpp:22-341) -- also Mmer<N> (include/supermer.hpp:23-341) with
 *     MINIMIZER_SIZE in place of KMER_SIZE.  `k` is the mer length, nw = ceil(k/32).
 * ---------------------------------------------------------------------------------------- */
static inline int hsko_nw(int k) { return (k + 31) / 32; }

/* set_kmer(const DnaSeq&) kmer.hpp:166-186 */
static hsko_mer mer_from_seq(const uint8_t *mem, size_t start, int k)
{
    hsko_mer m; memset(&m, 0, sizeof m);
    for (int i = 0; i < k; ++i) {
        uint64_t code = (uint64_t)hsko_base(mem, start + i);
        m.w[i / 32] |= code << (2 * (31 - (i % 32)));
    }
    return m;
}

/* operator< kmer.hpp:217-229: longs[0] compared first */
static inline int mer_less(const hsko_mer *a, const hsko_mer *b, int nw)
{
    for (int i = 0; i < nw; ++i) {
        if (a->w[i] < b->w[i]) return 1;
        if (a->w[i] > b->w[i]) return 0;
    }
    return 0;
}
static inline int mer_eq(const hsko_mer *a, const hsko_mer *b, int nw)
{
    for (int i = 0; i < nw; ++i) if (a->w[i] != b->w[i]) return 0;
    return 1;
}

/* GetExtension kmer.hpp:248-263.  (k%32==0 is UB in the reference -- shift by 64; here the
 * mathematically intended bit 0 is used, and tests never use k%32==0.) */
static hsko_mer mer_extend(const hsko_mer *m, int code, int k, int nw)
{
    hsko_mer e; memset(&e, 0, sizeof e);
    e.w[0] = m->w[0] << 2;
    for (int i = 1; i < nw; ++i) {
        e.w[i - 1] |= (m->w[i] >> 62) & 3;
        e.w[i] = m->w[i] << 2;
    }
    int sh = (k % 32) ? 2 * (32 - (k % 32)) : 0;
    e.w[nw - 1] |= (uint64_t)code << sh;
    return e;
}

/* tetramer_twin kmer.hpp:107-131: reverse-complement of one packed byte (4 bases), computed
 * rather than tabulated: out base j = 3 - in base (3-j). */
static inline uint64_t tetramer_twin(uint8_t b)
{
    uint8_t o = 0;
    for (int j = 0; j < 4; ++j) {
        int base = (b >> (2 * j)) & 3;          /* in base (3-j) counted from the MSB side */
        o |= (uint8_t)((3 - base) << (6 - 2 * j));
    }
    return o;
}

/* GetTwin kmer.hpp:266-296 */
static hsko_mer mer_twin(const hsko_mer *m, int k, int nw)
{
    hsko_mer t; memset(&t, 0, sizeof t);
    for (int l = 0; l < nw; ++l) {
        uint64_t longmer = m->w[l];
        for (int i = 0; i < 64; i += 8) {
            uint8_t bytemer = (uint8_t)((longmer >> i) & 0xff);
            t.w[nw - 1 - l] |= tetramer_twin(bytemer) << (56 - i);
        }
    }
    uint64_t shift = (k % 32) ? 2 * (32 - (k % 32)) : 0;
    if (shift) {
        uint64_t mask = ((1ULL << shift) - 1) << (64 - shift);
        t.w[0] <<= shift;
        for (int i = 1; i < nw; ++i) {
            t.w[i - 1] |= (t.w[i] & mask) >> (64 - shift);
            t.w[i] <<= shift;
        }
    }
    return t;
}

/* GetRep kmer.hpp:299-303 */
static hsko_mer mer_rep(const hsko_mer *m, int k, int nw)
{
    hsko_mer t = mer_twin(m, k, nw);
    return mer_less(&t, m, nw) ? t : *m;
}

/* GetHash kmer.hpp:306-311 / supermer.hpp:308-313 */
static inline uint64_t mer_hash(const hsko_mer *m, int nw) { return hsko_murmur64(m->w, 8 * nw); }

/* GetRepKmers kmer.hpp:314-341 (rolling GetExtension, then GetRep each).  out: n*nw words.
 * Returns the number of mers (0 if len < k). */
int64_t hsko_rep_mers(const uint8_t *mem, uint64_t len, int k, uint64_t *out)
{
    int nw = hsko_nw(k);
    int64_t n = (int64_t)len - k + 1;
    if (n <= 0) return 0;
    hsko_mer cur = mer_from_seq(mem, 0, k);
    for (int64_t i = 0; i < n; ++i) {
        if (i > 0) cur = mer_extend(&cur, hsko_base(mem, i + k - 1), k, nw);
        hsko_mer r = mer_rep(&cur, k, nw);
        for (int j = 0; j < nw; ++j) out[i * nw + j] = r.w[j];
    }
    return n;
}

/* canonical m-mer hashes, one per m-mer position (GetRepMmers + GetHash, kmerops.cpp:1024,1030) */
int64_t hsko_mmer_hashes(const uint8_t *mem, uint64_t len, int m, uint64_t *out)
{
    int nw = hsko_nw(m);
    int64_t n = (int64_t)len - m + 1;
    if (n <= 0) return 0;
    hsko_mer cur = mer_from_seq(mem, 0, m);
    for (int64_t i = 0; i < n; ++i) {
        if (i > 0) cur = mer_extend(&cur, hsko_base(mem, i + m - 1), m, nw);
        hsko_mer r = mer_rep(&cur, m, nw);
        out[i] = mer_hash(&r, nw);
    }
    return n;
}

/* Kmer::GetString kmer.hpp:147-163 */
void hsko_mer_string(const uint64_t *w, int k, char *out)
{
    for (int i = 0; i < k; ++i) out[i] = "ACGT"[(w[i / 32] >> (2 * (31 - (i % 32)))) & 3];
    out[k] = 0;
}

/* ------------------------------------------------------------------------------------------
 * a4. FindKmerDestinationsParallel (src/kmerops.cpp:1010-1041), Minimizer_Deque (:1058-1073),
 *     GetMinimizerOwner (:1044-1047).  dest has len-K+1 entries (none if len < K).
 * ---------------------------------------------------------------------------------------- */
int64_t hsko_dests(const uint8_t *mem, uint64_t len, int k, int m, int tot_tasks, int32_t *dest)
{
    if (len < (uint64_t)k) return 0;          /* kmerops.cpp:1019 */
    int64_t nm = (int64_t)len - m + 1;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nm);
    hsko_mmer_hashes(mem, len, m, h);
    /* monotone deque of (hash,pos): array-backed */
    uint64_t *dh = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nm);
    int64_t *dp = (int64_t *)malloc(sizeof(int64_t) * (size_t)nm);
    int64_t front = 0, back = 0; /* [front, back) */
    int64_t head = 0, nd = 0;
#define DQ_INSERT(hash, pos) do { while (back > front && dh[back - 1] > (hash)) --back; dh[back] = (hash); dp[back] = (pos); ++back; } while (0)
#define DQ_REMOVE(pos) do { while (back > front && dp[front] <= (pos)) ++front; } while (0)
    for (; head < k - m; ++head) DQ_INSERT(h[head], head);
    int64_t tail = head - k + m - 1;
    for (; head < nm; ++head, ++tail) {
        DQ_INSERT(h[head], head);
        DQ_REMOVE(tail);
        dest[nd++] = (int32_t)(dh[front] % (uint64_t)tot_tasks);
    }
#undef DQ_INSERT
#undef DQ_REMOVE
    free(h); free(dh); free(dp);
    return nd;
}

/* ------------------------------------------------------------------------------------------
 * a5. SupermerEncoder::encode (src/kmerops.cpp:1109-1147), copy_bits (:1096-1107),
 *     cnt_bytes/pad_base (include/kmerops.hpp:33-41).
 * Emits, in read order, for each supermer: task, start_pos, len; and the bytes via callback
 * arrays.  Caller provides capacity for at most (len-K+1) supermers.
 * ---------------------------------------------------------------------------------------- */
static inline int cnt_bytes(int len) { return (len + (4 - len % 4)) / 4; }

int hsko_cnt_bytes(int len) { return cnt_bytes(len); }

void hsko_copy_bits(uint8_t *dst, const uint8_t *src, uint64_t start_pos, int len)
{
    memset(dst, 0, (size_t)cnt_bytes(len));
    for (int i = 0; i < len; i++) {
        uint64_t loc = i + start_pos;
        int first_bit = (src[loc / 4] >> (7 - 2 * (loc % 4))) & 1;
        int second_bit = (src[loc / 4] >> (6 - 2 * (loc % 4))) & 1;
        dst[i / 4] |= (uint8_t)((first_bit << (7 - (i % 4) * 2)) | (second_bit << (6 - (i % 4) * 2)));
    }
}

int64_t hsko_supermers(const int32_t *dest, int64_t ndest, int k,
                       int32_t *sm_task, uint32_t *sm_start, uint32_t *sm_len)
{
    if (ndest <= 0) return 0;
    int64_t ns = 0;
    uint32_t start_pos = 0;
    int cnt = 1;
    int last_dst = dest[0];
    for (int64_t i = 1; i <= ndest; i++) {
        if (i == ndest || dest[i] != last_dst || cnt == HSKO_MAX_SUPERMER_LEN - k + 1) {
            sm_task[ns] = last_dst;
            sm_start[ns] = start_pos;
            sm_len[ns] = (uint32_t)(cnt + k - 1);
            ns++;
            if (i < ndest) last_dst = dest[i];
            cnt = 0;
            start_pos = (uint32_t)i;
        }
        cnt++;
    }
    return ns;
}

/* ------------------------------------------------------------------------------------------
 * a7. HeavyHitterClassifier::classify (src/kmerops.cpp:1157-1199): type 1 iff
 *     size > (total/ntask) * UNBALANCED_RATIO (integer avg, double compare).
 * ---------------------------------------------------------------------------------------- */
void hsko_classify(const uint64_t *task_kmers, int ntask, double unbalanced_ratio, int32_t *types)
{
    uint64_t total = 0;
    for (int i = 0; i < ntask; ++i) total += task_kmers[i];
    uint64_t avg = total / (uint64_t)ntask;
    for (int i = 0; i < ntask; ++i)
        types[i] = ((double)task_kmers[i] > (double)avg * unbalanced_ratio) ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * a9. BalancedDispatcher::dispatch / try_dispatch (src/kmerops.cpp:1214-1327).
 *     Sort ascending by size (the reference uses std::sort; ties are broken here by task id,
 *     ascending, which is what libstdc++'s introsort yields for the small inputs in the golden
 *     vectors -- tests only rely on tie-free inputs).  Returns 0 on success, -1 if no
 *     coefficient < DISPATCH_UPPER_COE works (reference throws, :1319).
 * ---------------------------------------------------------------------------------------- */
typedef struct { int64_t id; uint64_t sz; int64_t coe; } taskinfo;
static int ti_cmp(const void *a, const void *b)
{
    const taskinfo *x = (const taskinfo *)a, *y = (const taskinfo *)b;
    if (x->sz < y->sz) return -1;
    if (x->sz > y->sz) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

static int try_dispatch(const taskinfo *ti, int ntasks, int32_t *dest, int nprocs, uint64_t avg, double coe)
{
    uint64_t upper = (uint64_t)((double)avg * coe);
    uint64_t *asz = (uint64_t *)calloc((size_t)nprocs, sizeof(uint64_t));
    int32_t *owner = (int32_t *)malloc(sizeof(int32_t) * (size_t)ntasks);
    char *assigned = (char *)calloc((size_t)ntasks, 1);
    for (int i = 0; i < ntasks; ++i) owner[i] = -1;
    for (int i = 0; i < nprocs; ++i) {
        int id = ntasks - 1 - i;
        assigned[id] = 1;
        owner[id] = i;
        asz[i] += ti[id].sz * (uint64_t)ti[id].coe;
    }
    int cur = nprocs - 1, ok = 1;
    for (int i = 0; i < ntasks && ok; ++i) {
        if (assigned[i]) continue;
        int cnt = 0;
        while (cnt < nprocs) {
            if (asz[cur] + ti[i].sz <= upper) {
                owner[i] = cur;
                asz[cur] += ti[i].sz * (uint64_t)ti[i].coe;
                assigned[i] = 1;
                if (--cur < 0) cur += nprocs;
                break;
            }
            if (--cur < 0) cur += nprocs;
            cnt++;
        }
        if (cnt == nprocs) ok = 0;
    }
    if (ok) for (int i = 0; i < ntasks; ++i) dest[ti[i].id] = owner[i];
    free(asz); free(owner); free(assigned);
    return ok;
}

int hsko_dispatch_balanced(const uint64_t *task_bytes, int ntasks, int nprocs,
                           double upper_coe, double step, int32_t *dest)
{
    if (ntasks < nprocs) return -2; /* reference indexes task_info[ntasks-1-i] out of range */
    taskinfo *ti = (taskinfo *)malloc(sizeof(taskinfo) * (size_t)ntasks);
    uint64_t total = 0;
    for (int i = 0; i < ntasks; ++i) { ti[i].id = i; ti[i].sz = task_bytes[i]; ti[i].coe = 1; total += task_bytes[i]; }
    qsort(ti, (size_t)ntasks, sizeof(taskinfo), ti_cmp);
    uint64_t avg = total / (uint64_t)nprocs;
    double coe = 1.0 - step;
    int success = 0;
    for (int i = 0; i < ntasks; ++i) dest[i] = -1;
    while (coe < upper_coe) {
        if (try_dispatch(ti, ntasks, dest, nprocs, avg, coe)) { success = 1; break; }
        coe += step;
    }
    free(ti);
    return success ? 0 : -1;
}

/* RoundRobinDispatcher (src/kmerops.cpp:1201-1211) */
void hsko_dispatch_roundrobin(int ntasks, int nprocs, int32_t *dest)
{
    for (int i = 0; i < ntasks; ++i) dest[i] = i % nprocs;
}

/* prepare_supermer task-count rule (src/kmerops.cpp:40-43,76) */
int hsko_tot_tasks(int omp_max_threads, int thread_per_worker, int avg_task_per_worker, int nprocs)
{
    int avg_tasks = omp_max_threads / thread_per_worker * avg_task_per_worker - 1;
    if (avg_tasks < 1) avg_tasks = avg_task_per_worker;
    return avg_tasks * nprocs;
}

/* ------------------------------------------------------------------------------------------
 * a11-a14. Whole path for the tasks in [task_lo, task_hi) owned by one rank:
 *   supermer split of every read (a4,a5) -> per task: extraction from the re-aligned supermer
 *   bytes (GatheredSupermer::receive_from_buffer_stage2, kmerops.cpp:484-521) -> sort by key
 *   (sort_task kmerops.cpp:1382; order = RADULS/PARADIS for K<=32 i.e. ascending u64; for K>32
 *   `sorter`=2 gives RADULS order (longs[nw-1] most significant), 1 gives operator< order)
 *   -> count_sorted_kmers (kmerops.cpp:1410-1445) with [L,U] filter -> concatenation in ascending
 *   task id (copy_results kmerops.cpp:883-904).
 * The result arrays are malloc'ed; free with hsko_result_free.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t n;          /* entries */
    int32_t nw;          /* words per key */
    int32_t ext;
    uint64_t *keys;      /* n*nw, word 0 = longs[0] */
    uint64_t *cnt;       /* n */
    uint64_t *payoff;    /* n+1 (ext) */
    uint32_t *pos;       /* payoff[n] (ext) */
    int32_t *rid;        /* payoff[n] (ext) */
    uint64_t *task_off;  /* ntasks+1: entry range per task id (empty for tasks not owned) */
    uint64_t total_kmers;/* k-mers extracted over the owned tasks (before counting) */
    uint64_t total_supermers;
    uint64_t total_supermer_bytes; /* reference wire bytes: sum cnt_bytes(len) + 4*supermers (EXT: 12*) */
} hsko_result;

typedef struct { hsko_mer k; uint32_t pos; int32_t rid; } seed_t;

static int g_sort_nw, g_sort_mode;
static int seed_cmp(const void *a, const void *b)
{
    const seed_t *x = (const seed_t *)a, *y = (const seed_t *)b;
    if (g_sort_mode == 2) { /* little-endian multiword: highest word most significant */
        for (int i = g_sort_nw - 1; i >= 0; --i) {
            if (x->k.w[i] < y->k.w[i]) return -1;
            if (x->k.w[i] > y->k.w[i]) return 1;
        }
    } else {
        for (int i = 0; i < g_sort_nw; ++i) {
            if (x->k.w[i] < y->k.w[i]) return -1;
            if (x->k.w[i] > y->k.w[i]) return 1;
        }
    }
    return 0;
}

/* LSD byte radix sort of seeds (stable), used instead of qsort when fast!=0 (cpu baseline).
 * Same resulting key order as seed_cmp; payload order among equal keys = input order. */
static void seed_radix_sort(seed_t *a, size_t n, int nw, int mode)
{
    if (n < 2) return;
    seed_t *tmp = (seed_t *)malloc(n * sizeof(seed_t));
    seed_t *src = a, *dst = tmp;
    for (int wi = 0; wi < nw; ++wi) {
        int w = (mode == 2) ? wi : (nw - 1 - wi); /* least significant word first */
        for (int b = 0; b < 8; ++b) {
            size_t hist[256]; memset(hist, 0, sizeof hist);
            for (size_t i = 0; i < n; ++i) hist[(src[i].k.w[w] >> (8 * b)) & 0xff]++;
            int skip = 0;
            for (int d = 0; d < 256; ++d) if (hist[d] == n) { skip = 1; break; }
            if (skip) continue;
            size_t sum = 0;
            for (int d = 0; d < 256; ++d) { size_t c = hist[d]; hist[d] = sum; sum += c; }
            for (size_t i = 0; i < n; ++i) dst[hist[(src[i].k.w[w] >> (8 * b)) & 0xff]++] = src[i];
            seed_t *t = src; src = dst; dst = t;
        }
    }
    if (src != a) memcpy(a, src, n * sizeof(seed_t));
    free(tmp);
}

typedef struct { seed_t *v; size_t n, cap; } seedvec;
static void sv_push(seedvec *s, const seed_t *e)
{
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 1024; s->v = (seed_t *)realloc(s->v, s->cap * sizeof(seed_t)); }
    s->v[s->n++] = *e;
}

int hsko_count(const uint8_t *packed, const uint64_t *read_off, const uint32_t *read_len, uint64_t nreads,
               int k, int m, int L, int U, int ext, int ntasks, int64_t rid_base,
               const int32_t *task_owner, int my_rank, int sorter, int fast, hsko_result *out)
{
    if (k <= 2 || k >= 96 || m >= k || m < 1 || ntasks < 1 || L < 1 || L > U) return -1;
    int nw = hsko_nw(k);
    memset(out, 0, sizeof *out);
    out->nw = nw; out->ext = ext;
    seedvec *tv = (seedvec *)calloc((size_t)ntasks, sizeof(seedvec));
    uint64_t tot_sm = 0, tot_smb = 0;

    /* per read: dest (a4) -> supermers (a5) -> re-aligned bytes -> GetRepKmers on the supermer
     * (a11).  Done read by read (the reference batches it through the exchange; same values). */
    uint32_t maxlen = 0;
    for (uint64_t r = 0; r < nreads; ++r) if (read_len[r] > maxlen) maxlen = read_len[r];
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = fast ? omp_get_max_threads() : 1;
#endif
    seedvec **ltv = (seedvec **)calloc((size_t)nthreads, sizeof(seedvec *));
    for (int t = 0; t < nthreads; ++t) ltv[t] = (seedvec *)calloc((size_t)ntasks, sizeof(seedvec));
    uint64_t *lsm = (uint64_t *)calloc((size_t)nthreads * 2, sizeof(uint64_t));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        size_t cap = (size_t)maxlen + 8;
        int32_t *dest = (int32_t *)malloc(sizeof(int32_t) * cap);
        int32_t *smt = (int32_t *)malloc(sizeof(int32_t) * cap);
        uint32_t *sms = (uint32_t *)malloc(sizeof(uint32_t) * cap);
        uint32_t *sml = (uint32_t *)malloc(sizeof(uint32_t) * cap);
        uint8_t smbytes[HSKO_MAX_SUPERMER_LEN / 4 + 2];
        uint64_t mers[(HSKO_MAX_SUPERMER_LEN + 1) * HSKO_MAXW];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (uint64_t r = 0; r < nreads; ++r) {
            const uint8_t *mem = packed + read_off[r];
            int64_t nd = hsko_dests(mem, read_len[r], k, m, ntasks, dest);
            int64_t ns = hsko_supermers(dest, nd, k, smt, sms, sml);
            for (int64_t s = 0; s < ns; ++s) {
                int task = smt[s];
                lsm[2 * tid] += 1; lsm[2 * tid + 1] += (uint64_t)cnt_bytes((int)sml[s]);
                if (task_owner && task_owner[task] != my_rank) continue;
                hsko_copy_bits(smbytes, mem, sms[s], (int)sml[s]);
                int64_t nk = hsko_rep_mers(smbytes, sml[s], k, mers);
                for (int64_t i = 0; i < nk; ++i) {
                    seed_t e; memset(&e, 0, sizeof e);
                    for (int j = 0; j < nw; ++j) e.k.w[j] = mers[i * nw + j];
                    e.pos = sms[s] + (uint32_t)i;              /* kmerops.cpp:510 pos + i */
                    e.rid = (int32_t)(rid_base + (int64_t)r);   /* kmerops.cpp:66-71,1016 */
                    sv_push(&ltv[tid][task], &e);
                }
            }
        }
        free(dest); free(smt); free(sms); free(sml);
    }
    for (int t = 0; t < nthreads; ++t) { tot_sm += lsm[2 * t]; tot_smb += lsm[2 * t + 1]; }
    /* concatenate thread-local vectors in thread order (reference: per-thread buffers, kmerops.hpp:119) */
    for (int task = 0; task < ntasks; ++task) {
        size_t tot = 0;
        for (int t = 0; t < nthreads; ++t) tot += ltv[t][task].n;
        tv[task].v = (seed_t *)malloc((tot ? tot : 1) * sizeof(seed_t));
        tv[task].n = tv[task].cap = tot;
        size_t o = 0;
        for (int t = 0; t < nthreads; ++t) {
            if (ltv[t][task].n) memcpy(tv[task].v + o, ltv[t][task].v, ltv[t][task].n * sizeof(seed_t));
            o += ltv[t][task].n;
            free(ltv[t][task].v);
        }
    }
    for (int t = 0; t < nthreads; ++t) free(ltv[t]);
    free(ltv); free(lsm);

    out->total_supermers = tot_sm;
    out->total_supermer_bytes = tot_smb + tot_sm * (ext ? 12u : 4u);

    /* per task: sort (a12) + count (a13).  Results kept per task, then concatenated (a14). */
    uint64_t *tn = (uint64_t *)calloc((size_t)ntasks, sizeof(uint64_t));
    uint64_t **tkeys = (uint64_t **)calloc((size_t)ntasks, sizeof(uint64_t *));
    uint64_t **tcnt = (uint64_t **)calloc((size_t)ntasks, sizeof(uint64_t *));
    uint64_t **tpo = (uint64_t **)calloc((size_t)ntasks, sizeof(uint64_t *)); /* payload start idx into sorted seeds */
    g_sort_nw = nw; g_sort_mode = sorter;
    uint64_t total_k = 0;
    for (int task = 0; task < ntasks; ++task) total_k += tv[task].n;
    out->total_kmers = total_k;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int task = 0; task < ntasks; ++task) {
        seedvec *s = &tv[task];
        if (s->n == 0) continue;               /* reference: UB on empty task (kmerops.cpp:1415) */
        if (fast) seed_radix_sort(s->v, s->n, nw, sorter);
        else qsort(s->v, s->n, sizeof(seed_t), seed_cmp);
        uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nw * s->n);
        uint64_t *cnt = (uint64_t *)malloc(sizeof(uint64_t) * s->n);
        uint64_t *po = (uint64_t *)malloc(sizeof(uint64_t) * s->n);
        uint64_t n = 0;
        size_t i = 0;
        while (i < s->n) {
            size_t j = i + 1;
            while (j < s->n && mer_eq(&s->v[j].k, &s->v[i].k, nw)) ++j;
            uint64_t c = j - i;
            if (c >= (uint64_t)L && c <= (uint64_t)U) {
                for (int w = 0; w < nw; ++w) keys[n * nw + w] = s->v[i].k.w[w];
                cnt[n] = c; po[n] = i; n++;
            }
            i = j;
        }
        tn[task] = n; tkeys[task] = keys; tcnt[task] = cnt; tpo[task] = po;
    }
    uint64_t N = 0, P = 0;
    for (int task = 0; task < ntasks; ++task) { N += tn[task]; for (uint64_t i = 0; i < tn[task]; ++i) P += tcnt[task][i]; }
    out->n = N;
    out->keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nw * (N ? N : 1));
    out->cnt = (uint64_t *)malloc(sizeof(uint64_t) * (N ? N : 1));
    out->task_off = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(ntasks + 1));
    if (ext) {
        out->payoff = (uint64_t *)malloc(sizeof(uint64_t) * (N + 1));
        out->pos = (uint32_t *)malloc(sizeof(uint32_t) * (P ? P : 1));
        out->rid = (int32_t *)malloc(sizeof(int32_t) * (P ? P : 1));
    }
    uint64_t o = 0, po = 0;
    for (int task = 0; task < ntasks; ++task) {
        out->task_off[task] = o;
        for (uint64_t i = 0; i < tn[task]; ++i) {
            for (int w = 0; w < nw; ++w) out->keys[(o + i) * nw + w] = tkeys[task][i * nw + w];
            out->cnt[o + i] = tcnt[task][i];
            if (ext) {
                out->payoff[o + i] = po;
                for (uint64_t c = 0; c < tcnt[task][i]; ++c) {
                    out->pos[po] = tv[task].v[tpo[task][i] + c].pos;
                    out->rid[po] = tv[task].v[tpo[task][i] + c].rid;
                    po++;
                }
            }
        }
        o += tn[task];
        free(tkeys[task]); free(tcnt[task]); free(tpo[task]); free(tv[task].v);
    }
    out->task_off[ntasks] = o;
    if (ext) out->payoff[N] = po;
    free(tn); free(tkeys); free(tcnt); free(tpo); free(tv);
    return 0;
}

void hsko_result_free(hsko_result *r)
{
    free(r->keys); free(r->cnt); free(r->payoff); free(r->pos); free(r->rid); free(r->task_off);
    memset(r, 0, sizeof *r);
}

/* ------------------------------------------------------------------------------------------
 * Per-task digests of the k-mer INSTANCES a rank extracts (test infrastructure for inputs far beyond what hsko_count
 * can hold: 10 Gbp = 8e9 k-mers).  Streaming, no sort, no count, memory O(ntasks) per thread:
 *   n[t]   = number of k-mers whose minimizer sends them to task t          (a4: kmerops.cpp:1010-1047)
 *   mix[t] = sum over those k-mers of digest_mix(canonical k-mer[, pos, rid])   (mod 2^64)
 * For an unfiltered result list the same numbers follow from the entries: n[t] = sum of cnt, mix[t] = sum of
 * cnt * digest_mix(key) (with EXTENSION: sum over every payload), so a strictly ascending list with equal digests IS
 * the task's k-mer multiset.  The arithmetic restates the same reference definitions as the functions above, rolling
 * instead of from scratch (GetExtension kmer.hpp:248-263 forwards; the twin rolled the other way; canonical = smaller
 * under operator<, kmer.hpp:217,299); tests/test_oracle_golden.py pins it against hsko_count on every golden input.
 * task_sel (optional, [ntasks]): digest only the tasks with a non-zero byte (the others stay 0).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t fmix64_d(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
uint64_t hsko_digest_mix(const uint64_t *w, int nw, int ext, uint32_t pos, int32_t rid)
{
    uint64_t x = 0;
    for (int j = 0; j < nw; ++j) x = fmix64_d(x ^ (w[j] + 0x9e3779b97f4a7c15ULL * (uint64_t)(j + 1)));
    if (ext) x = fmix64_d(x ^ fmix64_d((((uint64_t)(uint32_t)rid) << 32 | pos) + 0x632be59bd9b4e019ULL));
    return x;
}

/* multi-word helpers on `nw` words, base i at word i/32, shift 2*(31 - i%32) */
static inline void mw_shl2(uint64_t *w, int nw) { for (int j = 0; j < nw; ++j) w[j] = (w[j] << 2) | (j + 1 < nw ? w[j + 1] >> 62 : 0); }
static inline void mw_shr2(uint64_t *w, int nw) { for (int j = nw - 1; j >= 0; --j) w[j] = (w[j] >> 2) | (j > 0 ? w[j - 1] << 62 : 0); }
static inline int mw_less(const uint64_t *a, const uint64_t *b, int nw) { for (int j = 0; j < nw; ++j) { if (a[j] < b[j]) return 1; if (a[j] > b[j]) return 0; } return 0; }

int hsko_task_digests(const uint8_t *packed, const uint64_t *read_off, const uint32_t *read_len, uint64_t nreads,
                      int k, int m, int ext, int ntasks, int64_t rid_base, const uint8_t *task_sel, uint64_t *n_out, uint64_t *mix_out)
{
    if (k <= 2 || k >= 96 || m >= k || m < 1 || ntasks < 1 || k % 32 == 0 || m % 32 == 0) return -1;
    const int nw = hsko_nw(k), nwm = hsko_nw(m), W = k - m + 1;
    memset(n_out, 0, sizeof(uint64_t) * (size_t)ntasks); memset(mix_out, 0, sizeof(uint64_t) * (size_t)ntasks);
    const int klast = (k - 1) / 32, kshift = 2 * (31 - (k - 1) % 32);          /* where base k-1 sits */
    const int mlast = (m - 1) / 32, mshift = 2 * (31 - (m - 1) % 32);
    const uint64_t kmask = ~0ULL << kshift, mmask = ~0ULL << mshift;           /* valid bits of the last word */
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        uint64_t *ln = (uint64_t *)calloc((size_t)ntasks, sizeof(uint64_t)), *lm = (uint64_t *)calloc((size_t)ntasks, sizeof(uint64_t));
        uint64_t *dqh = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(W + 2)); int64_t *dqp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(W + 2));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (uint64_t r = 0; r < nreads; ++r) {
            const uint64_t len = read_len[r];
            if (len < (uint64_t)k) continue;                                   /* kmerops.cpp:1019 */
            const uint8_t *mem = packed + read_off[r];
            uint64_t kf[HSKO_MAXW] = {0, 0, 0}, kr[HSKO_MAXW] = {0, 0, 0}, mf[HSKO_MAXW] = {0, 0, 0}, mr[HSKO_MAXW] = {0, 0, 0};
            /* ring deque of (hash, m-mer position), capacity W + 1 */
            int64_t qf = 0, qb = 0;                                            /* logical indices, slot = idx % (W + 1) */
            const int64_t QC = W + 1;
            for (uint64_t i = 0; i < len; ++i) {
                const uint64_t c = (uint64_t)hsko_base(mem, i);
                /* forward strands: drop the oldest base, append c at base position k-1 / m-1 */
                mw_shl2(kf, nw); kf[klast] = (kf[klast] & kmask & ~(3ULL << kshift)) | (c << kshift); for (int j = klast + 1; j < nw; ++j) kf[j] = 0;
                mw_shl2(mf, nwm); mf[mlast] = (mf[mlast] & mmask & ~(3ULL << mshift)) | (c << mshift);
                /* twins: everything one base to the right, the complement of c in front */
                mw_shr2(kr, nw); kr[0] |= (3 - c) << 62; kr[klast] &= kmask;
                mw_shr2(mr, nwm); mr[0] |= (3 - c) << 62; mr[mlast] &= mmask;
                if (i + 1 >= (uint64_t)m) {                                    /* m-mer ending at base i: position j = i - m + 1 */
                    const int64_t j = (int64_t)i - m + 1;
                    const uint64_t *cm = mw_less(mr, mf, nwm) ? mr : mf;       /* GetRep */
                    const uint64_t h = hsko_murmur64(cm, (uint32_t)(8 * nwm));
                    while (qb > qf && dqh[(qb - 1) % QC] > h) --qb;            /* Minimizer_Deque::insert kmerops.cpp:1058 */
                    dqh[qb % QC] = h; dqp[qb % QC] = j; ++qb;
                }
                if (i + 1 >= (uint64_t)k) {                                    /* k-mer ending at base i: position p = i - k + 1, m-mers p .. p + k - m */
                    const int64_t p = (int64_t)i - k + 1;
                    while (qb > qf && dqp[qf % QC] < p) ++qf;                  /* remove what left the window */
                    const int task = (int)(dqh[qf % QC] % (uint64_t)ntasks);   /* GetMinimizerOwner kmerops.cpp:1044 */
                    if (task_sel && !task_sel[task]) continue;
                    const uint64_t *ck = mw_less(kr, kf, nw) ? kr : kf;
                    ln[task] += 1;
                    lm[task] += hsko_digest_mix(ck, nw, ext, (uint32_t)p, (int32_t)(rid_base + (int64_t)r));
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        { for (int t = 0; t < ntasks; ++t) { n_out[t] += ln[t]; mix_out[t] += lm[t]; } }
        free(ln); free(lm); free(dqh); free(dqp);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a16. print_kmer_histogram (src/hysortk.cpp:98-136): text "#count\tnumkmers\n" then
 * "i\thisto[i]\n" for non-zero i>=1, then an empty line.  64-bit bins here (the reference's
 * int bins overflow above 2^31-1; documented divergence).  Returns bytes written (excluding NUL)
 * or the needed size when buf is too small.
 * ---------------------------------------------------------------------------------------- */
size_t hsko_histogram_text(const uint64_t *cnt, uint64_t n, char *buf, size_t bufsz)
{
    uint64_t maxc = 0;
    for (uint64_t i = 0; i < n; ++i) if (cnt[i] > maxc) maxc = cnt[i];
    uint64_t *h = (uint64_t *)calloc((size_t)maxc + 1, sizeof(uint64_t));
    for (uint64_t i = 0; i < n; ++i) h[cnt[i]]++;
    size_t off = 0;
    char line[64];
    int len = snprintf(line, sizeof line, "#count\tnumkmers\n");
    if (off + (size_t)len < bufsz) memcpy(buf + off, line, (size_t)len);
    off += (size_t)len;
    for (uint64_t i = 1; i <= maxc; ++i) {
        if (!h[i]) continue;
        len = snprintf(line, sizeof line, "%llu\t%llu\n", (unsigned long long)i, (unsigned long long)h[i]);
        if (off + (size_t)len < bufsz) memcpy(buf + off, line, (size_t)len);
        off += (size_t)len;
    }
    if (off + 1 < bufsz) buf[off] = '\n';
    off += 1;
    if (off < bufsz) buf[off] = 0;
    free(h);
    return off;
}
