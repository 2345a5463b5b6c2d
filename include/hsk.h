/*
 * hsk.h -- C ABI of libhsk.so, the MI355X (gfx950) k-mer counting hot path.
 *
 * This is the drop-in boundary for the reference's `kmerops` path: everything that
 * hysortk::kmer_count() (reference src/hysortk.cpp:36-96) does between receiving the
 * 2-bit packed DnaBuffer and returning the filtered KmerListS --
 *     prepare_supermer   (reference src/kmerops.cpp:23-127)
 *     exchange_supermer  (reference src/kmerops.cpp:130-195)
 *     filter_kmer        (reference src/kmerops.cpp:198-250)
 * -- runs behind hsk_count() as hand-written HIP kernels.  The C++ shim in
 * include/hysortk/ (same names and layouts as the reference's public headers) and the Python
 * host mirror (hysortk_amd/) both sit on top of exactly these entry points.
 *
 * Rules: plain C, POD only, no exceptions cross the boundary, every function returns an
 * hsk_status (0 = ok).  Buffers returned in hsk_result are owned by the library and released
 * by hsk_result_free().  One hsk_ctx per process and GPU; a ctx is not thread-safe.
 */
#ifndef HSK_H_
#define HSK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSK_ABI_VERSION 4

typedef enum {
    HSK_OK = 0,
    HSK_ERR_INVALID_ARG = 1,   /* bad K/M/L/U/ntasks or NULL pointer (reference: static_assert, compiletime.h:10-22) */
    HSK_ERR_NO_DEVICE = 2,     /* no HIP device / wrong arch: the product path never falls back to a CPU */
    HSK_ERR_HIP = 3,           /* a HIP runtime call failed; see hsk_last_error() */
    HSK_ERR_OOM = 4,           /* device or pinned-host allocation failed */
    HSK_ERR_DISPATCH = 5,      /* "Cannot dispatch tasks. May be too unbalanced." (reference kmerops.cpp:1319) */
    HSK_ERR_INTERNAL = 6,      /* device-side consistency check failed (e.g. look-back timeout) */
    HSK_ERR_COMM = 7,          /* RCCL not available / collective failed */
    HSK_ERR_UNSUPPORTED = 8    /* valid in the reference, not built yet here */
} hsk_status;

/* Runtime form of the reference's compile-time macros (reference Makefile:1-46). */
typedef struct {
    int32_t kmer_size;        /* KMER_SIZE        2 < K < 96, K % 32 != 0 (reference UB, kmer.hpp:260) */
    int32_t minimizer_size;   /* MINIMIZER_SIZE   0 < M < K, M % 32 != 0 (one-, two- and three-word minimizers) */
    int32_t lower_freq;       /* LOWER_KMER_FREQ  1 <= L <= U */
    int32_t upper_freq;       /* UPPER_KMER_FREQ  U <= 65535 */
    int32_t extension;        /* EXTENSION        0 | 1: carry (PosInRead, ReadId) through the sort */
    int32_t ntasks;           /* tot_tasks of prepare_supermer (kmerops.cpp:40-43,76); 0 = pick for the GPU */
    int32_t device;           /* HIP device ordinal */
    int32_t plain_dispatcher; /* PLAIN_DISPATCHER: 1 = round robin, 0 = balanced (kmerops.cpp:115-119) */
    double  dispatch_upper_coe; /* DISPATCH_UPPER_COE (1.5) */
    double  dispatch_step;      /* DISPATCH_STEP      (0.05) */
    int32_t radix_bits;       /* digit width of the LSD radix sort: 8 (default) */
    int32_t flags;            /* HSK_FLAG_* */
    double  unbalanced_ratio; /* UNBALANCED_RATIO (2.3): a task is a heavy hitter above ratio x the mean task (kmerops.cpp:1190); ABI 3 */
    const char *tuning;       /* ABI 4 (was reserved[0]): NULL, or "name=value,name=value": forced paths for tests and a few measured thresholds, read per
                                 context (rounds 1-3: process-wide environment variables).  Copied by hsk_init. */
    int64_t reserved[2];
} hsk_config;

#define HSK_FLAG_PROFILE      1   /* HIP-event timing of every radix scatter launch (hsk_stats) */
#define HSK_FLAG_KEEP_DEVICE  2   /* hsk_count_device leaves the result in HBM (entries_dev) */
#define HSK_FLAG_PLAIN_CLASSIFIER 4 /* PLAIN_CLASSIFIER: no heavy-hitter pre-aggregation (kmerops.cpp:109-113) */
/* Which algorithm orders and counts the k-mers of a task (ABI 3).  Default: two radix scatter passes on the top 16 key bits,
 * then one LDS hash aggregation per 16-bit prefix bin -- the fast plan for sequencing data, where every k-mer repeats about
 * `coverage` times; the library leaves it by itself when the input turns out to hold (nearly) only unique k-mers. */
#define HSK_FLAG_NO_AGGREGATION 8   /* never aggregate in LDS tables: four scatter passes on the top 32 bits + an in-LDS finish per
                                       tile (one-word keys), the full sort below for every other record shape */
#define HSK_FLAG_NO_COMBINE     32   /* never take the combining extraction (one GPU, one-word keys, no payload, from 64 MB of packed reads on:
                                       the supermers are ordered by minimizer bucket, every bucket's k-mers are counted in an LDS table where
                                       they are extracted, and only the {k-mer, count} pairs enter the passes above); the library leaves it by
                                       itself when the input yields more than one pair per sixteen k-mers (reads with ~0.2 % errors and more, coverage below ~20) */
#define HSK_FLAG_FULL_SORT     16   /* the reference's own algorithm: LSD radix sort over ALL key bytes of every task
                                       (sort_task, kmerops.cpp:1382) + adjacent-equal merge-count over the sorted array
                                       (count_sorted_kmers, kmerops.cpp:1410); nothing fused, nothing skipped */

typedef struct hsk_ctx hsk_ctx;

/*
 * Result of one hsk_count(): the rank-local KmerListS.
 *   entries: n records of (nw + 1) uint64 = { TKmer longs[nw]; uint64_t cnt } -- byte-identical
 *            to the reference's KmerListEntryS for EXTENSION == 0 (kmer.hpp:368-383), so the
 *            C++ shim memcpy's it into std::vector<KmerListEntryS>.
 *   order:   ascending task id over the tasks this rank owns (copy_results, kmerops.cpp:883-904);
 *            inside a task ascending as a little-endian multi-word integer (= sort_task,
 *            kmerops.cpp:1382, RADULS order; for K <= 32 this is plain ascending uint64).
 *   EXTENSION == 1: entry i owns pos/rid[payload_off[i] .. payload_off[i] + cnt_i): its slice of the
 *            task's sorted payload array (slices of filtered-out k-mers stay in pos/rid as unused
 *            gaps; payload_off[n] = total payload length).
 */
typedef struct {
    uint64_t n;               /* number of (k-mer, count) entries */
    int32_t  nw;              /* 64-bit words per k-mer: 1 (K<=32), 2 (K<=64), 3 (K<=95) */
    int32_t  ntasks;          /* tot_tasks actually used */
    uint64_t *entries;        /* host (pinned) n * (nw+1) words; NULL with HSK_FLAG_KEEP_DEVICE */
    uint64_t *task_off;       /* host, ntasks+1: entry range of each task id (empty if not owned) */
    uint64_t *payload_off;    /* host, n+1 (EXTENSION): first payload of entry i; [n] = length of pos/rid */
    uint32_t *pos;            /* host, payload_off[n] values (EXTENSION): PosInRead */
    int32_t  *rid;            /* host, payload_off[n] values (EXTENSION): ReadId */
    uint64_t *histo;          /* host, histo_len bins: histo[c] = #entries with cnt == c */
    uint64_t histo_len;
    void     *entries_dev;    /* device copy when HSK_FLAG_KEEP_DEVICE (owned by the ctx) */
    /* --- measurements of this call --- */
    uint64_t total_kmers;     /* k-mer instances of this rank's tasks (after the exchange; a heavy-hitter task's, which arrive as lists, incl.),
                                 plus the instances this rank's scan left out as certain to be dropped (hsk_stats::dropped_kmers): over the
                                 ranks they add up to the k-mers of the input */
    uint64_t total_supermers;
    uint64_t total_supermer_bytes;
    double   ms_total;        /* device time of the whole path (HIP events) */
    double   ms_parse;        /* minimizer/supermer kernels */
    double   ms_exchange;     /* RCCL all-to-all (0 on one GPU) */
    double   ms_extract;      /* supermer -> canonical k-mer kernels */
    double   ms_sort;         /* histogram + all radix passes */
    double   ms_count;        /* merge-count + filter */
    double   ms_d2h;          /* result copy to the host */
    void    *priv;            /* library bookkeeping */
} hsk_result;

/* Per-kernel accounting (HIP events on the launch stream around every launch of the named kernels), filled when
 * HSK_FLAG_PROFILE is set.  bytes = algorithmic bytes (records read + written). */
typedef struct {
    uint64_t scatter_launches;
    uint64_t scatter_keys;        /* sum over launches of keys moved */
    uint64_t scatter_bytes;       /* sum over launches of 2 * record_bytes * keys */
    double   scatter_ms;          /* sum of launch durations (HIP events on the launch stream) */
    uint64_t hist_launches;
    uint64_t hist_bytes;
    double   hist_ms;
    int64_t  fused_tasks;         /* tasks finished by a fused finish kernel (aggregating or tile finish) */
    int64_t  redone_tasks;        /* tasks the hybrid path had to redo with the full-width passes */
    uint64_t agg_launches;        /* aggregating finish kernel (hsk_agg.h): launches, record bytes read, duration */
    uint64_t agg_bytes;
    double   agg_ms;
    int64_t  agg_retried_tasks;   /* tasks that needed the large hash table (a bin with many distinct keys) */
    int64_t  parse_fallbacks;     /* parses that left the fast path (a tile with more supermers than the record capacity) */
    int64_t  heavy_tasks;         /* heavy-hitter tasks this rank pre-aggregated and shipped as k-mer lists (multi-GPU) */
    int64_t  dropped_kmers;       /* k-mer instances the scan left out because their k-mer -- the all-A or the all-C k-mer -- has more than U copies inside the
                                     plan's sample alone and so cannot be in the result (round 4; the field was `onepass_misses`, always 0, before) */
    /* --- ABI 2 --- */
    uint64_t scan_launches;       /* scan_kernel (minimizers + supermer records): launches, packed read bytes, duration */
    uint64_t scan_bytes;
    double   scan_ms;
    uint64_t place_launches;      /* place_kernel (supermers to their task slots): launches, supermers placed, duration */
    uint64_t place_supermers;
    double   place_ms;
    uint64_t host_syncs;          /* blocking waits of the host on the GPU (stream or event) inside hsk_count* since the last reset */
    uint64_t host_waits_covered;  /* ... of which the host had already enqueued the next batch's kernels behind the awaited work,
                                     so the wait does not leave the GPU idle */
    uint64_t h2d_bytes;           /* hsk_count(): input bytes copied (or read in place over PCIe) from the host */
    uint64_t d2h_bytes;           /* result bytes copied to the host */
    double   h2d_ms;              /* duration of the input copies (0 when the parse reads pinned host memory in place) */
    double   d2h_ms;              /* duration of the result copies, whether or not they overlapped the kernels */
    /* --- ABI 3, combining extraction (hsk_combine.h) --- */
    uint64_t bucket_launches;     /* bucket order of the supermer items (histogram + scatter launches), items moved, duration */
    uint64_t bucket_items;
    double   bucket_ms;
    uint64_t combine_launches;    /* combine_kernel: launches, k-mers counted in LDS tables, {k-mer, count} pairs written, duration */
    uint64_t combine_kmers;
    uint64_t combine_pairs;
    double   combine_ms;
} hsk_stats;

/* ---- lifecycle ------------------------------------------------------------------------- */
int  hsk_abi_version(void);
int  hsk_device_count(void);                       /* HIP devices visible to this process (0 if none / no driver) */
/* Pinned (page-locked) host memory for the DnaBuffer handed to hsk_count(): such a buffer is read by the GPU in place, the
 * transfer overlaps the minimizer scan.  Pageable memory works too (staged copies first).  NULL when the allocation fails. */
void *hsk_host_alloc(uint64_t bytes);
void  hsk_host_free(void *p);
int  hsk_init(const hsk_config *cfg, hsk_ctx **out);
void hsk_destroy(hsk_ctx *ctx);
const char *hsk_strerror(int status);
const char *hsk_last_error(const hsk_ctx *ctx);   /* detail text of the last failure */
void hsk_config_default(hsk_config *cfg);          /* reference Makefile defaults: K=31 M=17 L=15 U=40 */

/* ---- the hot path ---------------------------------------------------------------------- */
/*
 * hsk_count: replaces prepare_supermer + exchange_supermer + filter_kmer for the reads of
 * this rank.  Input is the reference's DnaBuffer memory as is (dnabuffer.hpp:14, dnaseq.hpp:33):
 *   packed        2-bit bases, 4 per byte, first base in the two MSBs; every read starts on a
 *                 byte boundary
 *   read_byte_off nreads offsets into packed (DnaBuffer::getbufoffset)
 *   read_len      nreads lengths in bases (DnaSeq::size)
 *   rid_base      global id of read 0 (reference: MPI_Exscan of read counts, kmerops.cpp:65-71)
 * Collective over the communicator when hsk_comm_init() has been called.
 */
int hsk_count(hsk_ctx *ctx, const uint8_t *packed, uint64_t packed_bytes,
              const uint64_t *read_byte_off, const uint32_t *read_len, uint64_t nreads,
              int64_t rid_base, hsk_result *out);

/* Same, inputs already resident in HBM (d_read_byte_off has nreads entries). */
int hsk_count_device(hsk_ctx *ctx, const void *d_packed, uint64_t packed_bytes,
                     const void *d_read_byte_off, const void *d_read_len, uint64_t nreads,
                     int64_t rid_base, hsk_result *out);

/* Test/diagnostic entry: `nranks` VIRTUAL ranks on the one GPU of this ctx.  Runs the multi-GPU data path
 * (task-size probe, dispatch, owner-grouped parse, byte packing, all-to-all-v plan, multi-segment extraction)
 * with device-to-device copies in place of the RCCL send/recv; outs[r] is what rank r would return.
 * owner_out (optional, capacity in entries) receives the task -> rank table. */
int hsk_count_loopback(hsk_ctx *ctx, int nranks, const uint8_t *const *packed, const uint64_t *packed_bytes,
                       const uint64_t *const *read_byte_off, const uint32_t *const *read_len, const uint64_t *nreads,
                       hsk_result *outs, int32_t *owner_out, int32_t owner_capacity);

/* The same with every virtual rank's reads already resident in HBM (d_read_byte_off[r] has nreads[r] entries): full-size runs of the
 * multi-rank data path on one GPU.  outs[r].ms_parse / ms_exchange / ms_extract / ms_sort / ms_count / ms_total are rank r's device
 * times (the virtual ranks run one after the other). */
int hsk_count_loopback_device(hsk_ctx *ctx, int nranks, const void *const *d_packed, const uint64_t *packed_bytes,
                              const void *const *d_read_byte_off, const void *const *d_read_len, const uint64_t *nreads,
                              hsk_result *outs, int32_t *owner_out, int32_t owner_capacity);

/* ctx == NULL is allowed when the context is already destroyed: a result outlives its context (its pinned host blocks belong to the
 * result; hsk_destroy frees only the context's cache of released blocks). */
void hsk_result_free(hsk_ctx *ctx, hsk_result *res);
/* HSK_FLAG_KEEP_DEVICE: where task `task`'s share of the result lives in HBM -- entries (n records of nw + 1 words), and with
 * EXTENSION the CSR payload: payload_off (n values in the rank's payload numbering), pos / rid (npay values each, the
 * task's whole sorted payload) and payload_base: entry i owns pos / rid[payload_off[i] - payload_base ...][0 .. cnt_i).
 * A downstream GPU stage (e.g. an overlap detector consuming (ReadId, PosInRead) lists, reference README.md:52,74) reads
 * them in place; the memory belongs to the result. */
int hsk_result_device_task(const hsk_result *res, int32_t task, const void **entries, uint64_t *n,
                           const void **payload_off, const void **pos, const void **rid, uint64_t *npay, uint64_t *payload_base);
/* Result egress (write_output_file, reference src/hysortk.cpp:138-164): the lines "KMERSTRING\tcount\n" of `n` entries
 * ((nw + 1)-word records: host memory, or device memory when `on_device`), formatted on the GPU into `text` (host, capacity
 * bytes); *nbytes = bytes needed (call with capacity 0 to size the buffer: K + 2 + up to 20 digits per entry). */
int hsk_format_entries(hsk_ctx *ctx, const void *entries, uint64_t n, int32_t nw, int32_t on_device,
                       char *text, uint64_t capacity, uint64_t *nbytes);
int  hsk_get_stats(hsk_ctx *ctx, hsk_stats *out, int reset);

/* ---- stage entry points (each one is a reference function of SURVEY 8a; used by the parity
 *      tests and by callers that want one stage only) ---------------------------------------- */
/* a4: FindKmerDestinationsParallel (kmerops.cpp:1010): dest[j] for every k-mer of every read,
 * concatenated in read order; dest_off[r] .. dest_off[r+1] is read r.  Host in / host out. */
int hsk_stage_destinations(hsk_ctx *ctx, const uint8_t *packed, uint64_t packed_bytes,
                           const uint64_t *read_byte_off, const uint32_t *read_len, uint64_t nreads,
                           int32_t *dest, uint64_t dest_capacity, uint64_t *dest_off);
/* a5+a11: supermer split followed by GetRepKmers: the canonical k-mers (nw words each, plus
 * pos,rid when EXTENSION) of task `task`, in unspecified order.  Returns count in *n. */
int hsk_stage_task_kmers(hsk_ctx *ctx, const uint8_t *packed, uint64_t packed_bytes,
                         const uint64_t *read_byte_off, const uint32_t *read_len, uint64_t nreads,
                         int64_t rid_base, int32_t task, uint64_t *keys, uint32_t *pos, int32_t *rid,
                         uint64_t capacity, uint64_t *n);
/* a12: sort_task (kmerops.cpp:1382): in-place LSD radix sort of n records of nw words by their
 * first key_bytes bytes read as a little-endian integer; optional 8-byte payload per record. */
int hsk_stage_sort(hsk_ctx *ctx, uint64_t *keys, uint64_t *vals, uint64_t n, int32_t nw);
/* a13: count_sorted_kmers (kmerops.cpp:1410) on a sorted host array; writes entries
 * {key words, cnt} to out_entries (capacity in entries), *n_out = number kept. */
int hsk_stage_count_sorted(hsk_ctx *ctx, const uint64_t *sorted_keys, uint64_t n, int32_t nw,
                           uint64_t *out_entries, uint64_t capacity, uint64_t *n_out);

/* ---- host-side planning (pure CPU; usable without a GPU) ---------------------------------- */
/* tot_tasks rule of prepare_supermer (kmerops.cpp:40-43,76). */
int hsk_plan_tot_tasks(int omp_max_threads, int thread_per_worker, int avg_task_per_worker, int nprocs);
/* HeavyHitterClassifier (kmerops.cpp:1157-1199). */
int hsk_plan_classify(const uint64_t *task_kmers, int ntasks, double unbalanced_ratio, int32_t *types);
/* BalancedDispatcher / RoundRobinDispatcher (kmerops.cpp:1201-1327): task -> owner rank. */
int hsk_plan_dispatch(const uint64_t *task_bytes, int ntasks, int nprocs, int plain,
                      double upper_coe, double step, int32_t *owner);
/* FastaIndex::getpartition (fastaindex.cpp:52-100): contiguous split of reads by bases. */
int hsk_plan_partition_reads(const uint64_t *read_len, uint64_t nreads, int nprocs, uint64_t *counts);

/* The all-to-all-v plan of the supermer exchange (csrc/hsk_comm.h), as pure arithmetic: given the
 * owner table and the full size matrix M[src][task] = {supermers, bytes, kmers}, what `rank` sends
 * to / receives from each peer (counts and offsets in supermers and bytes; a rank's tasks are stored
 * grouped by owner, ascending task id) and where every (task, src) segment lands in the receive
 * arrays.  send_recv: [nranks][8] = send_sup, send_bytes, send_sup_off, send_byte_off, recv_sup,
 * recv_bytes, recv_sup_off, recv_byte_off.  segs: [ntasks][nranks][4] = sup_off, n_sup, byte_off,
 * kmer_off (zero rows for tasks `rank` does not own). */
int hsk_plan_exchange(int nranks, int rank, int ntasks, const int32_t *owner, const uint64_t *size_matrix,
                      uint64_t *send_recv, uint64_t *segs);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) -------------------------------------- */
#define HSK_UNIQUE_ID_BYTES 128
int hsk_comm_get_unique_id(void *id128);                       /* rank 0, then broadcast by the caller */
int hsk_comm_init(hsk_ctx *ctx, int nranks, int rank, const void *id128);
int hsk_comm_destroy(hsk_ctx *ctx);
/* One-rank communicator on this GPU: all-reduce and grouped send/recv to self with checked results (exercises the
 * RCCL binding on a single-GPU box). */
int hsk_comm_selftest(hsk_ctx *ctx);

/* ---- synthetic reads, generated in HBM (bench / tests) ------------------------------------ */
/* S-reads(G, c) of BASELINE.md: random genome of genome_len bases, error-free reads of read_len
 * bases sampled uniformly, strand 50/50.  Outputs device pointers owned by the ctx (freed by
 * hsk_synth_free or hsk_destroy).  The same generator exists in numpy (hysortk_amd/synth.py). */
int hsk_synth_reads(hsk_ctx *ctx, uint64_t genome_len, uint32_t read_len, uint64_t nreads, uint64_t seed,
                    uint64_t first_read, /* index of read 0 in the global read stream (rank * nreads for weak scaling) */
                    void **d_packed, uint64_t *packed_bytes, void **d_read_byte_off, void **d_read_len);
/* Same with substitution errors: every base is replaced by one of the other three with probability error_rate (0 .. 0.75; at 0.75 every base is uniform over ACGT).
 * Error-free reads are the best case of the LDS hash aggregation (distinct keys per prefix bin = records / coverage); with
 * ~1 % errors most bins hold several times more distinct keys. */
int hsk_synth_reads_err(hsk_ctx *ctx, uint64_t genome_len, uint32_t read_len, uint64_t nreads, uint64_t seed, uint64_t first_read,
                        double error_rate, void **d_packed, uint64_t *packed_bytes, void **d_read_byte_off, void **d_read_len);
int hsk_synth_free(hsk_ctx *ctx, void *d_packed, void *d_read_byte_off, void *d_read_len);

/* ---- FASTA ingest on the device (what read_dna_buffer + DnaSeq::compress do on the host, reference src/hysortk.cpp:18-33,
 *      src/dnaseq.cpp:9-31) -------------------------------------------------------------------------------------------- */
/* `text` = the FASTA file's bytes (host memory, e.g. an mmap); record r has rec_len[r] bases, the first at text[rec_pos[r]],
 * with line_bases[r] bases per line and line_width[r] bytes per line including the line break (the .fai columns 2-5;
 * line_bases 0 = no line breaks).  Builds the 2-bit packed DnaBuffer of these records in HBM (every record byte-aligned,
 * same bytes as the reference's packer incl. its code-4 spill for non-ACGTN characters) and returns device arrays that
 * hsk_count_device takes (free them with hsk_synth_free). */
int hsk_pack_fasta(hsk_ctx *ctx, const char *text, uint64_t text_bytes, const uint64_t *rec_pos, const uint32_t *rec_len,
                   const uint32_t *line_bases, const uint32_t *line_width, uint64_t nrec,
                   void **d_packed, uint64_t *packed_bytes, void **d_read_byte_off, void **d_read_len);
int hsk_memcpy_d2h(hsk_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);
/* Measurement aid (ABI 3): what a plain HBM copy reaches on this GPU -- a hand-written kernel, 16 bytes per lane, `bytes` read and
 * `bytes` written per launch, best of `iters` launches over four grid sizes; *gbs = 2 * bytes / time.  bench.py quotes the
 * kernels' achieved GB/s against the 8 TB/s spec AND against this figure. */
int hsk_copy_peak(hsk_ctx *ctx, uint64_t bytes, int iters, double *gbs);

#ifdef __cplusplus
}
#endif
#endif /* HSK_H_ */
