// hsk_host_parse.h -- host side of the parse stage: supermer store, parse_count / parse_place (kernels: hsk_parse.h).
// Part of the single translation unit hsk_api.hip (included in this order; everything here is file-local).
#pragma once

// ------------------------------------------------------------------------------------------------
// stage: parse (a4, a5, a6)
// ------------------------------------------------------------------------------------------------
// where the bases of a task's supermers live
struct BaseSource {
    const u64 *src8 = nullptr; u64 bit0 = 0; u64 nwords = 0;   // byte stream rounded down to 8 bytes
    const u64 *gpos = nullptr;                                  // reference mode positions (null: prefix-sum byte offsets)
    const u32 *boff = nullptr;                                  // byte-store mode: supermer s starts at byte seg.byte_off + boff[s] of the stream
    const void *bitems = nullptr; u32 n_bitems = 0;             // item mode with scan-placed bins: the bucket order's work list (BucketItem[], one per chunk), built on the device
    const u32 *sub = nullptr; const u64 *item = nullptr;        // item mode (combining extraction): minimizer bits and the two item words of every supermer (SupermerStore::sm_sub / sm_item); nothing else
};
static bool reads_in_place(const BaseSource &b) { return b.gpos != nullptr || b.boff != nullptr; }   // no prefix sums needed to find a supermer's bases
static BaseSource source_from_packed(const u8 *d_packed, u64 packed_bytes, const u64 *gpos)
{
    BaseSource b; const uintptr_t p = (uintptr_t)d_packed;
    b.src8 = (const u64 *)(p & ~(uintptr_t)7); b.bit0 = 8 * (u64)(p & 7); b.nwords = ((p & 7) + packed_bytes + 7) / 8; b.gpos = gpos;
    return b;
}
static BaseSource source_from_bytes(const u8 *bytes, u64 nbytes)
{
    BaseSource b; b.src8 = (const u64 *)bytes; b.bit0 = 0; b.nwords = (nbytes + 7) / 8 + 1; b.gpos = nullptr;   // pool blocks are padded
    return b;
}

struct SupermerStore {
    u32 ntasks = 0, nblocks = 0;
    u8 *sm_len = nullptr; u8 *sm_bytes = nullptr; u64 *sm_gpos = nullptr; u32 *sm_pos = nullptr; int32_t *sm_rid = nullptr;
    u32 *sm_boff = nullptr;       // byte-store mode (place_bytes_kernel): sm_bytes is complete, sm_boff[slot] = offset inside the task's byte run
    void *d_bitems = nullptr; u32 n_bitems = 0; void *bin_aux[4] = {nullptr, nullptr, nullptr, nullptr};      // scan-placed bins: work list + cursor / map / control / chunk owners (released with the store)
    bool bytes_done = false;      // place_bytes_kernel has written sm_bytes (supermers that travel: without sm_boff -- the receiver scans the lengths anyway)
    unsigned short *sm_sub16 = nullptr;   // several ranks, combining extraction on the owner's side: the top 16 minimizer bits of every supermer (they travel with sm_len)
    u32 *sm_sub = nullptr;        // combining extraction (hsk_combine.h): 32 mixed bits of the supermer's minimizer hash ...
    u64 *sm_item = nullptr;       // ... and the supermer itself (place_items_kernel); sm_len / sm_gpos do not exist in this mode
    u64 tot_sup = 0, tot_bytes = 0, tot_kmers = 0;
    std::vector<u64> task_tot;    // [ntasks][3] supermers, bytes, kmers
    std::vector<u64> task_base;   // [ntasks][3] slot, byte, kmer bases (tasks stored in `order`)
    std::vector<u32> order;       // storage order of tasks (grouped by owner rank, ascending id)
    BaseSource base;              // multi-GPU: where pack_kernel reads the bases from (the rank's packed reads)
    std::vector<char> group_packed;  // multi-GPU, grouped exchange: sm_bytes of task group g have been produced
};

static void free_store(hsk_ctx *c, SupermerStore &s)
{
    c->pool.release(s.sm_sub16); s.sm_sub16 = nullptr;
    c->pool.release(s.d_bitems); s.d_bitems = nullptr; s.n_bitems = 0; for (void *&p : s.bin_aux) { c->pool.release(p); p = nullptr; }
    c->pool.release(s.sm_len); c->pool.release(s.sm_bytes); c->pool.release(s.sm_gpos); c->pool.release(s.sm_pos); c->pool.release(s.sm_rid); c->pool.release(s.sm_boff); c->pool.release(s.sm_sub); s.sm_sub = nullptr; c->pool.release(s.sm_item); s.sm_item = nullptr;
    s.sm_len = s.sm_bytes = nullptr; s.sm_gpos = nullptr; s.sm_pos = nullptr; s.sm_rid = nullptr; s.sm_boff = nullptr;
}

// where the extraction finds the bases of the store's supermers: the store's own byte runs, or (position mode) the packed reads
static BaseSource source_from_store(const SupermerStore &st, const u8 *d_packed, u64 packed_bytes)
{
    if (!st.sm_boff) { BaseSource b = source_from_packed(d_packed, packed_bytes, st.sm_gpos); b.sub = st.sm_sub; b.item = st.sm_item; b.bitems = st.d_bitems; b.n_bitems = st.n_bitems; return b; }
    BaseSource b = source_from_bytes(st.sm_bytes, st.tot_bytes);
    b.boff = st.sm_boff;
    return b;
}

// Byte-store placement or position placement?  Measured on 10 Gbp (one GPU): the byte store cuts the extraction's fetch
// traffic from ~6.5x to ~1x its algorithmic read bytes, but the extraction does not get faster (38.3 against 37.6 ms: it is
// bound by its chain of dependent phases, not by bandwidth) while the placement pays for the extra stores (12.8 against
// 8.4 ms).  With several GPUs the bytes must be produced anyway (they are what travels), and writing them here replaces
// pack_kernel's scattered gather (~20 ms per rank and step).  So: byte store when the supermers travel, positions when
// they stay; HSK_PLACE_BYTES=0/1 forces either.
static bool place_bytes_enabled(bool supermers_travel)
{
    const int env = (int)tune("place_bytes", -1);
    return env < 0 ? supermers_travel : env != 0;
}

// nslabs > 1 (scan_kernel / place_kernel / place_bytes_kernel only): every workgroup owns one run of tiles per slab, see ParseArgs
static ParseArgs make_parse_args(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen,
                                 u64 nreads, int64_t rid_base, u32 ntasks, u32 *nblocks_out, u32 nslabs = 1)
{
    ParseArgs a; memset(&a, 0, sizeof a);
    a.packed = d_packed; a.packed_bytes = packed_bytes; a.roff = d_roff; a.rlen = d_rlen; a.nreads = nreads;
    a.k = c->cfg.kmer_size; a.m = c->cfg.minimizer_size; a.ntasks = ntasks; a.fm = make_fastmod(ntasks >> c->vt_shift); a.vt_shift = c->vt_shift;      // (virtual tasks: `ntasks` counts them)
    a.item_maxk = (u32)std::max(1, std::min(16, 61 - c->cfg.kmer_size));
    a.ntiles = (packed_bytes * 4 + PARSE_TILE - 1) / PARSE_TILE;
    u32 nblocks = (u32)std::min<u64>(a.ntiles, c->scan_blocks ? c->scan_blocks : 1024);
    a.rid_base = rid_base;
    if (nslabs > 1 && a.ntiles >= (u64)nblocks * nslabs) {
        const u64 per_slab = (a.ntiles + nslabs - 1) / nslabs;
        a.tiles_per_block = (u32)((per_slab + nblocks - 1) / nblocks);
        a.slab_tiles = (u64)a.tiles_per_block * nblocks;
        a.nslabs = (u32)((a.ntiles + a.slab_tiles - 1) / a.slab_tiles);
        *nblocks_out = nblocks;
        return a;
    }
    a.tiles_per_block = (u32)((a.ntiles + nblocks - 1) / nblocks);
    nblocks = (u32)((a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block);
    a.nslabs = 1; a.slab = 0; a.slab_tiles = 0;
    *nblocks_out = nblocks;
    return a;
}

// The parse in two steps, so that a caller that needs the task sizes before it can fix the storage order
// (multi-GPU: sizes -> all-reduce -> dispatcher -> owner-grouped order) hashes the reads only once:
//   parse_count: minimizers + supermer boundaries of every tile; per-(workgroup, task) counts; task totals
//   parse_place: exclusive scan of the counts in the storage order `order`, supermers to their slots
// Fast path = scan_kernel + place_kernel (compact supermer records kept in between); the general path
// (M > 25, or a tile with more supermers than the record capacity) = parse_kernel<COUNT> + emit_kernel.
// scan_kernel places the items of the combining extraction itself (one GPU; ParseArgs::bin_*, hsk_parse.h: bin_place): bins = (XCD, virtual
// task) chunk lists.  The chunk store is sized from the EXPECTED supermer count (a window of W k-mers starts a supermer every (W + 1) / 2
// positions on random sequence, every 16 k-mers at least, once per read) with a quarter to spare; real genomes with long low-complexity
// stretches make FEWER supermers, and an input that does overrun it (error bit 128; a bin beyond its map: 256) sends the call round again
// without the combining extraction.  tuning "scan_place=1" selects it (default: tile records + place_items_kernel, see scan_place_enabled).
struct ScanBins {
    u32 *cursor = nullptr, *map = nullptr, *ctl = nullptr, *chunk_bin = nullptr, *subs = nullptr; ulonglong2 *items = nullptr;
    BinTable *d_table = nullptr;                          // the table scan_kernel<.., true> reads (ParseArgs::bins)
    u32 vmax = 0, cap = 0, nchunks = 0, nbins = 0;
};
// Measured (10 Gbp, round 4): scan 47.6 ms with the items placed by it against 33.9 + 13.4 ms (scan + place_items_kernel): the same ~13.5 ms wherever the
// placement runs, 101.7 ms per step either way -- and from host memory the separate placement hides behind the ingest's DMA while the longer scan
// does not (140 against 123 ms).  So the default stays the placement kernel; "scan_place=1" takes the fused form (no tile records: 13 GB less device
// memory and 20 GB less traffic per 10 Gbp).
static bool scan_place_enabled() { return tune("scan_place", 0) != 0; }
static void bins_release(hsk_ctx *c, ScanBins &b)
{
    c->pool.release(b.cursor); c->pool.release(b.map); c->pool.release(b.ctl); c->pool.release(b.chunk_bin); c->pool.release(b.subs); c->pool.release(b.items); c->pool.release(b.d_table);
    b = ScanBins();
}
static int bins_alloc(hsk_ctx *c, ScanBins &b, u32 nvt, u64 packed_bytes, u64 nreads, int W, hipStream_t stream)
{
    b = ScanBins();
    const double positions = (double)packed_bytes * 4.0;
    const double maxk = (double)std::max(1, std::min(16, 61 - c->cfg.kmer_size));      // k-mers per item at most (ParseArgs::item_maxk)
    const double expect = positions * std::max(2.2 / (double)(W + 1), 1.2 / maxk) + 2.0 * (double)nreads + 65536.0;
    b.nbins = 8u * nvt;
    if (b.nbins > 8192) return fail(c, HSK_ERR_UNSUPPORTED, "more than 8192 bins");      // (a chunk's owner word: 13 bits of bin)
    const u64 cap = (u64)(expect * (double)tune("bin_cap_pct", 125) / 100.0 / BIN_CHUNK) + (tune("bin_cap_pct", 125) >= 100 ? 2 * b.nbins + 64 : b.nbins + 1);      // (+ a first chunk and a chunk opened ahead per bin)      // (tests: a store that runs out)
    if (cap >= 0xFFFFFFF0ULL) return fail(c, HSK_ERR_UNSUPPORTED, "item store of %llu chunks", (unsigned long long)cap);
    b.cap = (u32)cap;
    b.vmax = (u32)std::min<u64>(std::max<u64>(cap / b.nbins * 64, 256), 1u << 16);
    if (tune("bin_vmax", 0) > 0) b.vmax = (u32)tune("bin_vmax", 0);                                    // (tests: a bin beyond its map)
    DALLOC(c, b.cursor, u32 *, (size_t)b.nbins * 4 * BIN_CUR_STRIDE);
    DALLOC(c, b.map, u32 *, (size_t)b.nbins * b.vmax * 4);
    DALLOC(c, b.ctl, u32 *, 256);
    DALLOC(c, b.chunk_bin, u32 *, (size_t)b.cap * 4 + 64);
    DALLOC(c, b.items, ulonglong2 *, (size_t)b.cap * BIN_CHUNK * 16 + 64);
    DALLOC(c, b.subs, u32 *, (size_t)b.cap * BIN_CHUNK * 4 + 64);
    HIPCHK(c, hipMemsetAsync(b.cursor, 0, (size_t)b.nbins * 4 * BIN_CUR_STRIDE, stream));
    HIPCHK(c, hipMemsetAsync(b.map, 0, (size_t)b.nbins * b.vmax * 4, stream));
    HIPCHK(c, hipMemsetAsync(b.ctl, 0, 256, stream));
    DALLOC(c, b.d_table, BinTable *, 256);
    BinTable *h = (BinTable *)((char *)c->pinned + (384u << 10));      // (pinned staging: the copy is asynchronous)
    h->cursor = b.cursor; h->map = b.map; h->vmax = b.vmax; h->cap_chunks = b.cap; h->ctl = b.ctl; h->chunk_bin = b.chunk_bin; h->items = b.items; h->subs = b.subs; h->err = c->d_err;
    HIPCHK(c, hipMemcpyAsync(b.d_table, h, sizeof(BinTable), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(bins_init_kernel, dim3((b.nbins + 255) / 256), dim3(256), 0, stream, *h, b.nbins);
    return HSK_OK;
}
static void bins_args(hsk_ctx *, ParseArgs &a, const ScanBins &b) { a.bins = b.d_table; }
// the store takes the bins over: items and minimizer bits where they are, the bucket order's work list from the chunk lists
static int bins_to_store(hsk_ctx *c, ScanBins &b, SupermerStore &st, u32 nvt, u32 vt_shift, hipStream_t stream);

constexpr u64 SLAB_HEAD = 4096;                    // bytes at the start of an ingest slab that the scan of the slab before it reads (>= (PARSE_WORDS - 128) * 4 + the prefetch's reach)
struct ParseJob {
    ParseArgs a; u32 nblocks = 0, ntasks = 0; bool fast = false, empty = true;
    const u64 *d_roff = nullptr; u64 nreads = 0; int64_t rid_base = 0;
    u64 *d_blk_cnt = nullptr; u16 *d_dest_cache = nullptr; u32 *d_tile_rec = nullptr, *d_tile_nrec = nullptr, *d_overflow = nullptr;
    u32 *d_tile_r0 = nullptr;     // EXTENSION: first read of every tile (hint for the (pos, rid) lookup)
    u32 *d_tile_sub = nullptr;    // combining extraction: minimizer bits of every record
    unsigned long long *d_dropped = nullptr;      // scan_kernel<.., DROP>: positions left out (hsk_ctx::drop_mask_now)
    ScanBins bins;                // ... or no records at all: scan_kernel places the items itself
    std::vector<u64> task_tot;    // [ntasks][3] supermers, bytes, kmers of this rank
};

static void parse_release(hsk_ctx *c, ParseJob &j)
{
    c->pool.release(j.d_blk_cnt); c->pool.release(j.d_dest_cache); c->pool.release(j.d_tile_rec); c->pool.release(j.d_tile_nrec); c->pool.release(j.d_overflow);
    c->pool.release(j.d_tile_r0); j.d_tile_r0 = nullptr; c->pool.release(j.d_tile_sub); j.d_tile_sub = nullptr;
    c->pool.release(j.d_dropped); j.d_dropped = nullptr;
    bins_release(c, j.bins);
    j.d_blk_cnt = nullptr; j.d_dest_cache = nullptr; j.d_tile_rec = j.d_tile_nrec = j.d_overflow = nullptr;
}

// place_items_kernel stages 16384 records + 32 tiles of packed words: ~158 KB of dynamic LDS, which the runtime wants announced
static size_t place_items_lds(u32 ntasks)
{
    static bool announced = false;
    if (!announced) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(place_items_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); announced = true; }
    return (size_t)ntasks * 16 + 8 + (size_t)PLACE_ITEM_REC * 8 + (size_t)PLACE_ITEM_WORDS * 4;
}
// the scan of the COUNT matrix: parse_scan_kernel (one workgroup) for the usual few dozen tasks, the three-kernel form from 128 columns on
static int launch_parse_scan(hsk_ctx *c, hipStream_t st, const u64 *blk_cnt, u32 nblocks, u32 ntasks, const u32 *d_order, const u8 *d_skip,
                             u64 *d_task_tot, u64 *d_task_base, u64 *d_blk_base, u64 *d_run = nullptr, u64 *d_scratch = nullptr)
{
    if (ntasks < 128) {
        hipLaunchKernelGGL(parse_scan_kernel, dim3(1), dim3(HSK_MAX_TASKS), 0, st, blk_cnt, nblocks, ntasks, d_order, d_skip, d_task_tot, d_task_base, d_blk_base, d_run);
        return HSK_OK;
    }
    u64 *d_part = d_scratch;                              // [PS_SEGS][ntasks][3]
    if (!d_part) DALLOC(c, d_part, u64 *, (size_t)PS_SEGS * ntasks * 3 * 8);
    const dim3 grid((ntasks + PS_THREADS - 1) / PS_THREADS, PS_SEGS);
    hipLaunchKernelGGL(parse_scan_part_kernel, grid, dim3(PS_THREADS), 0, st, blk_cnt, nblocks, ntasks, d_skip, d_part);
    hipLaunchKernelGGL(parse_scan_base_kernel, dim3(1), dim3(HSK_MAX_TASKS), 0, st, (const u64 *)d_part, ntasks, d_order, d_task_tot, d_task_base, d_run);
    hipLaunchKernelGGL(parse_scan_fill_kernel, grid, dim3(PS_THREADS), 0, st, blk_cnt, nblocks, ntasks, d_skip, (const u64 *)d_part, (const u64 *)d_task_base, d_blk_base);
    if (!d_scratch) c->pool.release(d_part);              // (stream-ordered reuse: the caller's later work is enqueued on the same stream)
    return HSK_OK;
}
static bool parse_fast_enabled()
{
    return tune("parse_fast", 1) != 0;
}
// Record slots per 2048-position tile.  A window of W k-mers starts a supermer every (W + 1) / 2 positions on random
// sequence (+ one per 128 positions and per read): 40 % head room, a power of two from SCAN_REC_CAP (W >= 12) to 2048
// (tiny windows); a tile that still overflows sends the parse through the general kernels.
static u32 parse_rec_cap(int W)
{
    const int forced = tune("parse_rec_cap", 0) ? (int)std::min<long long>(std::max<long long>(tune("parse_rec_cap", 0), 1), (long long)PLACE_MAX_REC) : 0;
    if (forced) return (u32)forced;
    const u32 need = (u32)(1.4 * (2.0 * PARSE_TILE / (W + 1) + 40));
    u32 cap = SCAN_REC_CAP;
    while (cap < need && cap < 2048) cap <<= 1;
    return cap;
}

static int parse_count(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                       int64_t rid_base, u32 ntasks, ParseJob &j)
{
    j = ParseJob();
    j.ntasks = ntasks; j.d_roff = d_roff; j.nreads = nreads; j.rid_base = rid_base;
    j.task_tot.assign((size_t)ntasks * 3, 0);
    if (nreads == 0 || packed_bytes == 0) return HSK_OK;            // nothing to parse on this rank
    j.empty = false;
    j.fast = parse_fast_enabled() && c->cfg.minimizer_size <= SCAN_MAX_M;
    // slab ingest (hsk_count() from pinned memory): only the fast path hashes slab by slab; everything else wants the reads in HBM first
    const u8 *h2d_src = c->h2d_src; c->h2d_src = nullptr;
    if (h2d_src && !j.fast) { HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed), h2d_src, packed_bytes, hipMemcpyHostToDevice, c->stream)); h2d_src = nullptr; }
    j.a = make_parse_args(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, ntasks, &j.nblocks, h2d_src ? (u32)c->h2d_slabs : 1);
    ParseArgs &a = j.a;
    if (h2d_src && a.nslabs <= 1) { HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed), h2d_src, packed_bytes, hipMemcpyHostToDevice, c->stream)); h2d_src = nullptr; }   // (too small for slabs)
    DALLOC(c, j.d_blk_cnt, u64 *, (size_t)1024 * ntasks * 3 * 8);
    a.blk_cnt = j.d_blk_cnt;
    u64 *d_task_tot; DALLOC(c, d_task_tot, u64 *, (size_t)ntasks * 3 * 8 + 64);
    u32 *h_ovf = (u32 *)((char *)c->pinned + c->pinned_bytes - 320);
    *h_ovf = 0;
    if (j.fast) {
        a.rec_cap = parse_rec_cap(c->cfg.kmer_size - c->cfg.minimizer_size + 1);
        a.place_group = std::max<u32>(1, std::min<u32>(16, PLACE_MAX_REC / a.rec_cap));
        const bool bins = c->item_mode_now && c->combine_now && scan_place_enabled() && a.rec_cap <= PLACE_ITEM_REC;
        DALLOC(c, j.d_overflow, u32 *, 256);
        HIPCHK(c, hipMemsetAsync(j.d_overflow, 0, 4, c->stream));
        a.overflow = j.d_overflow;
        if (bins) {
            int brc = bins_alloc(c, j.bins, ntasks, packed_bytes, nreads, c->cfg.kmer_size - c->cfg.minimizer_size + 1, c->stream); if (brc) return brc;
            bins_args(c, a, j.bins);
        } else {
            DALLOC(c, j.d_tile_rec, u32 *, (size_t)a.ntiles * a.rec_cap * 4 + 64);
            DALLOC(c, j.d_tile_nrec, u32 *, (size_t)a.ntiles * 4 + 64);
            a.tile_rec = j.d_tile_rec; a.tile_nrec = j.d_tile_nrec;
        }
        if (c->cfg.extension && nreads < (1ULL << 32)) { DALLOC(c, j.d_tile_r0, u32 *, (size_t)a.ntiles * 4 + 64); a.tile_r0 = j.d_tile_r0; }
        if (!bins && c->combine_now && a.rec_cap <= PLACE_ITEM_REC) { DALLOC(c, j.d_tile_sub, u32 *, (size_t)a.ntiles * a.rec_cap * 4 + 64); a.tile_sub = j.d_tile_sub; }
        const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
        if (c->zc_src) { a.packed = c->zc_src; a.packed_copy = (u32 *)const_cast<u8 *>(d_packed); }      // ingest fused into the scan
        EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 3; ep.bytes = packed_bytes; (void)hipEventRecord(ep.a, c->stream); }
        const bool scan_generic = tune("scan_generic", 0) != 0;      // (tests: the default (k, m) through the generic instance)
        c->dropped_now = 0;
        a.drop_mask = bins ? 0u : c->drop_mask_now;
        if (a.drop_mask) { DALLOC(c, j.d_dropped, unsigned long long *, 256); HIPCHK(c, hipMemsetAsync(j.d_dropped, 0, 8, c->stream)); a.dropped = j.d_dropped; }
        auto launch_scan = [&]() {
            if (a.drop_mask) hipLaunchKernelGGL((scan_kernel<0, 0, false, true>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
            else if (a.bins) {
                if (a.k == 31 && a.m == 17 && !scan_generic) hipLaunchKernelGGL((scan_kernel<31, 17, true>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
                else hipLaunchKernelGGL((scan_kernel<0, 0, true>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
            }
            else if (a.k == 31 && a.m == 17 && !scan_generic) hipLaunchKernelGGL((scan_kernel<31, 17>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
            else if (a.k == 51 && a.m == 17 && !scan_generic) hipLaunchKernelGGL((scan_kernel<51, 17>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
            else hipLaunchKernelGGL((scan_kernel<0, 0>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        };
        if (h2d_src) {
            // The packed reads cross PCIe as DMA copies, slab by slab, on the copy stream; the scan of slab s is launched behind the
            // copy of slab s + 1 (a tile's windows reach a few words into the next tile), so the link and the VALUs work side by side:
            // the parse takes about as long as the transfer (a kernel reading the host buffer in place gets ~40 GB/s out of the link,
            // the DMA engine ~55).
            EvList evs(c);
            std::vector<hipEvent_t> landed(a.nslabs), head(a.nslabs);      // (head: the first SLAB_HEAD bytes of a slab -- all the scan of the slab before it needs of it)
            EvPair hp{}; if (profile) { hp.a = ev_get(c); hp.b = ev_get(c); hp.kind = 5; (void)hipEventRecord(hp.a, c->d2h_stream); }
            for (u32 sl = 0; sl < a.nslabs; ++sl) {
                const u64 b0 = std::min<u64>((u64)sl * a.slab_tiles * (PARSE_TILE / 4), packed_bytes), b1 = std::min<u64>(((u64)sl + 1) * a.slab_tiles * (PARSE_TILE / 4), packed_bytes);
                const u64 bh = std::min<u64>(b0 + SLAB_HEAD, b1);
                if (bh > b0) HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed) + b0, h2d_src + b0, bh - b0, hipMemcpyHostToDevice, c->d2h_stream));
                head[sl] = evs.get();
                HIPCHK(c, hipEventRecord(head[sl], c->d2h_stream));
                if (b1 > bh) HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed) + bh, h2d_src + bh, b1 - bh, hipMemcpyHostToDevice, c->d2h_stream));
                landed[sl] = evs.get();
                HIPCHK(c, hipEventRecord(landed[sl], c->d2h_stream));
            }
            if (profile) { (void)hipEventRecord(hp.b, c->d2h_stream); c->ev_pending.push_back(hp); }
            for (u32 sl = 0; sl < a.nslabs; ++sl) {
                HIPCHK(c, hipStreamWaitEvent(c->stream, landed[sl], 0));
                if (sl + 1 < a.nslabs) HIPCHK(c, hipStreamWaitEvent(c->stream, head[sl + 1], 0));
                a.slab = sl;
                launch_scan();
            }
            a.slab = 0;
        } else launch_scan();
        if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
        a.packed = d_packed; a.packed_copy = nullptr; c->zc_src = nullptr;                                  // everything after the scan reads the copy in HBM
        hipLaunchKernelGGL(task_totals_kernel, dim3(1), dim3(HSK_MAX_TASKS), 0, c->stream, j.d_blk_cnt, j.nblocks, ntasks, d_task_tot);
        HIPCHK(c, hipMemcpyAsync(h_ovf, j.d_overflow, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_ovf + 1, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
        if (j.bins.items) HIPCHK(c, hipMemcpyAsync(h_ovf + 2, j.bins.ctl, 4, hipMemcpyDeviceToHost, c->stream));
        if (j.d_dropped) HIPCHK(c, hipMemcpyAsync(h_ovf + 4, j.d_dropped, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(j.task_tot.data(), d_task_tot, (size_t)ntasks * 3 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hsk_sync(c, c->stream));
        if (j.d_dropped) c->dropped_now = *(const unsigned long long *)(h_ovf + 4);
        if (j.bins.items) j.bins.nchunks = std::min(h_ovf[2], j.bins.cap);
        bool gaps = c->roff_bad; c->roff_bad = false;
        if (c->roff_check.valid() && !c->roff_check.get()) gaps = true;
        if (gaps) {
            // the buffer's reads do not lie back to back (the host threads found a gap while the GPU scanned): everything again
            // with the caller's offsets, copied now and validated on the device
            u64 *given = c->roff_given;
            c->pool.release(d_task_tot);
            parse_release(c, j);
            if (c->rlen_host) {                                    // (the lengths were a guess from a sample: the real ones now)
                HIPCHK(c, hipMemcpyAsync(const_cast<u32 *>(d_rlen), c->rlen_host, nreads * 4, hipMemcpyHostToDevice, c->stream));
                c->stats.h2d_bytes += nreads * 4; c->rlen_host = nullptr;
            }
            HIPCHK(c, hipMemcpyAsync(given, c->roff_host, nreads * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(given + nreads, d_roff + nreads, 8, hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(index_check_kernel, dim3(1024), dim3(256), 0, c->stream, given, d_rlen, nreads, packed_bytes, c->d_err);
            c->index_unchecked = true;
            c->stats.h2d_bytes += nreads * 8;
            return parse_count(c, d_packed, packed_bytes, given, d_rlen, nreads, rid_base, ntasks, j);
        }
        if (c->index_unchecked) {
            c->index_unchecked = false;
            if (h_ovf[1] & 32u) { (void)hipMemsetAsync(c->d_err, 0, 4, c->stream); c->pool.release(d_task_tot); return fail(c, HSK_ERR_INVALID_ARG, "the read index is not ascending / overlaps / leaves the packed buffer"); }
        }
        if (j.bins.items && (h_ovf[1] & (2u | 128u | 256u))) {          // the chunk store or a bin's map ran out: the call again, without the combining extraction
            (void)hipMemsetAsync(c->d_err, 0, 4, c->stream); c->pool.release(d_task_tot); c->combine_veto = true; return retry_plan("the scan-placed items' chunk store or a bin's map ran out (error word)", h_ovf[1]);
        }
        if (*h_ovf && c->vt_shift) { c->pool.release(d_task_tot); c->combine_veto = true; return retry_plan("a tile beyond the record capacity"); }      // (the general kernels know no virtual tasks: the call again, without them)
        if (*h_ovf) {                                                // a tile with more supermers than the record capacity
            j.fast = false; c->stats.parse_fallbacks++;
            c->dropped_now = 0;                                       // (the general kernels count again; they honour the same mask: several ranks stay consistent)
            if (a.nslabs > 1) {                                       // the general kernels know one tile range per workgroup
                ParseArgs b = make_parse_args(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, ntasks, &j.nblocks);
                b.blk_cnt = a.blk_cnt; b.rec_cap = a.rec_cap; b.place_group = a.place_group;
                a = b;
            }
            c->pool.release(j.d_tile_r0); j.d_tile_r0 = nullptr; a.tile_r0 = nullptr;
            c->pool.release(j.d_tile_rec); c->pool.release(j.d_tile_nrec); j.d_tile_rec = j.d_tile_nrec = nullptr;
            a.tile_rec = a.tile_nrec = nullptr;
            c->pool.release(j.d_tile_sub); j.d_tile_sub = nullptr; a.tile_sub = nullptr;
        }
    }
    if (c->zc_src) {                                                  // the general kernels read the reads more than once: plain copy first
        HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed), c->zc_src, packed_bytes, hipMemcpyDefault, c->stream));
        c->zc_src = nullptr;
    }
    if (!j.fast) {
        // task id per base position, kept from COUNT to EMIT (2 B x 4 x packed_bytes); optional: without it EMIT re-hashes
        j.d_dest_cache = (u16 *)c->pool.alloc((size_t)a.ntiles * PARSE_TILE * 2);
        a.dest_cache = j.d_dest_cache;
        a.drop_mask = c->drop_mask_now;                               // (a fresh ParseArgs after a slab fallback has lost it)
        if (a.drop_mask) {
            if (!j.d_dropped) DALLOC(c, j.d_dropped, unsigned long long *, 256);
            HIPCHK(c, hipMemsetAsync(j.d_dropped, 0, 8, c->stream)); a.dropped = j.d_dropped;
        }
        if (a.drop_mask) hipLaunchKernelGGL((parse_kernel<PARSE_COUNT, false, true>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        else hipLaunchKernelGGL((parse_kernel<PARSE_COUNT, false>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        hipLaunchKernelGGL(task_totals_kernel, dim3(1), dim3(HSK_MAX_TASKS), 0, c->stream, j.d_blk_cnt, j.nblocks, ntasks, d_task_tot);
        HIPCHK(c, hipMemcpyAsync(h_ovf + 1, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
        if (a.drop_mask) HIPCHK(c, hipMemcpyAsync(h_ovf + 4, j.d_dropped, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(j.task_tot.data(), d_task_tot, (size_t)ntasks * 3 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hsk_sync(c, c->stream));
        if (a.drop_mask) c->dropped_now = *(const unsigned long long *)(h_ovf + 4);
        // (hsk_count derives the read index only for the fast parse; should a derived index ever arrive here with its verdict still open, a
        //  wrong guess must not be counted: the call fails instead of trusting lengths and offsets nobody has confirmed)
        if (c->roff_check.valid() && !c->roff_check.get()) { c->pool.release(d_task_tot); return fail(c, HSK_ERR_INTERNAL, "derived read index reached the general parse unverified and does not match the caller's"); }
        if (c->index_unchecked) {
            c->index_unchecked = false;
            if (h_ovf[1] & 32u) { (void)hipMemsetAsync(c->d_err, 0, 4, c->stream); c->pool.release(d_task_tot); return fail(c, HSK_ERR_INVALID_ARG, "the read index is not ascending / overlaps / leaves the packed buffer"); }
        }
    }
    HIPCHK(c, hipGetLastError());
    c->pool.release(d_task_tot);
    return HSK_OK;
}

// `order` (storage order of tasks) must be a permutation of 0..ntasks-1.
// skip (optional, [ntasks]): tasks whose supermers are not stored (they take no room and report zero totals)
static int parse_place(hsk_ctx *c, ParseJob &j, const std::vector<u32> &order, SupermerStore &st, const std::vector<u8> *skip = nullptr, bool supermers_travel = false)
{
    const bool ext = c->cfg.extension != 0;
    const u32 ntasks = j.ntasks;
    st = SupermerStore();
    st.ntasks = ntasks; st.nblocks = j.nblocks; st.order = order;
    st.task_tot = j.task_tot;
    if (skip) for (u32 t = 0; t < ntasks; ++t) if ((*skip)[t]) st.task_tot[3 * t] = st.task_tot[3 * t + 1] = st.task_tot[3 * t + 2] = 0;
    st.task_base.assign((size_t)ntasks * 3, 0);
    { u64 s = 0, b = 0, k = 0;
      for (u32 i = 0; i < ntasks; ++i) { const u32 t = order[i]; st.task_base[3 * t] = s; st.task_base[3 * t + 1] = b; st.task_base[3 * t + 2] = k;
                                          s += st.task_tot[3 * t]; b += st.task_tot[3 * t + 1]; k += st.task_tot[3 * t + 2]; }
      st.tot_sup = s; st.tot_bytes = b; st.tot_kmers = k; }
    if (j.empty) return HSK_OK;
    ParseArgs &a = j.a;
    u64 *d_blk_base, *d_task_tot, *d_task_base; u32 *d_order;
    DALLOC(c, d_blk_base, u64 *, (size_t)j.nblocks * ntasks * 2 * 8);
    DALLOC(c, d_task_tot, u64 *, (size_t)ntasks * 3 * 8);
    DALLOC(c, d_task_base, u64 *, (size_t)ntasks * 3 * 8);
    DALLOC(c, d_order, u32 *, (size_t)ntasks * 4);
    HIPCHK(c, hipMemcpyAsync(d_order, order.data(), (size_t)ntasks * 4, hipMemcpyHostToDevice, c->stream));
    u8 *d_skip = nullptr;
    if (skip) {
        DALLOC(c, d_skip, u8 *, ntasks);
        HIPCHK(c, hipMemcpyAsync(d_skip, skip->data(), ntasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hsk_sync(c, c->stream));          // the mask is host memory of the caller
    }
    a.task_skip = d_skip;
    { int prc = launch_parse_scan(c, c->stream, j.d_blk_cnt, j.nblocks, ntasks, d_order, d_skip, d_task_tot, d_task_base, d_blk_base); if (prc) return prc; }
    const bool from_bins = j.fast && j.bins.items != nullptr && !skip && !supermers_travel && !ext;      // scan_kernel has placed the items already
    const bool item_mode = from_bins || (j.fast && a.tile_sub && !skip && !supermers_travel && !ext);      // combining extraction: the slots hold the supermers themselves
    if (!item_mode) DALLOC(c, st.sm_len, u8 *, st.tot_sup + 64);
    // byte-store mode (fast parse path): the supermers' bases are copied into per-task byte runs while the reads stream through
    // place_bytes_kernel once; positions are kept only where something still needs them (EXTENSION: pos / rid lookup)
    bool bytes_mode = !item_mode && j.fast && place_bytes_enabled(supermers_travel) && a.rec_cap <= PLACE_BYTES_REC;
    for (u32 t = 0; t < ntasks && bytes_mode; ++t) if (st.task_tot[3 * t + 1] >= (1ULL << 32)) bytes_mode = false;     // 32-bit offsets inside a task's run
    // several ranks with the combining extraction planned (c->combine_now): the supermers' minimizer bits go into the store as well
    const bool with_sub16 = bytes_mode && supermers_travel && c->combine_now && a.tile_sub && !ext;      // (heavy-hitter tasks are skipped by the kernel itself: their k-mers travel as lists)
    a.sm_sub16 = nullptr;
    if (with_sub16) { DALLOC(c, st.sm_sub16, unsigned short *, st.tot_sup * 2 + 64); a.sm_sub16 = st.sm_sub16; }
    if (bytes_mode) {
        DALLOC(c, st.sm_bytes, u8 *, st.tot_bytes + 256);
        if (!supermers_travel) DALLOC(c, st.sm_boff, u32 *, st.tot_sup * 4 + 64);
        st.bytes_done = true;
        HIPCHK(c, hipMemsetAsync(st.sm_bytes + st.tot_bytes, 0, 256, c->stream));     // the extraction's windows read a few words past the last supermer
    }
    if ((!bytes_mode && !item_mode) || ext) DALLOC(c, st.sm_gpos, u64 *, st.tot_sup * 8 + 64);      // position mode: bases stay in the packed reads
    if (ext) { DALLOC(c, st.sm_pos, u32 *, st.tot_sup * 4 + 64); DALLOC(c, st.sm_rid, int32_t *, st.tot_sup * 4 + 64); }
    a.sm_sub = nullptr; a.sm_item = nullptr;
    if (from_bins) {
        int brc = bins_to_store(c, j.bins, st, ntasks, c->vt_shift, c->stream);
        c->pool.release(d_blk_base); c->pool.release(d_task_tot); c->pool.release(d_task_base); c->pool.release(d_order); c->pool.release(d_skip);
        a.task_skip = nullptr;
        return brc;
    }
    if (item_mode) { DALLOC(c, st.sm_sub, u32 *, st.tot_sup * 4 + 64); DALLOC(c, st.sm_item, u64 *, st.tot_sup * 16 + 64); a.sm_sub = st.sm_sub; a.sm_item = st.sm_item; }
    a.blk_base = d_blk_base; a.sm_len = st.sm_len; a.sm_gpos = st.sm_gpos; a.sm_bytes = st.sm_bytes; a.sm_boff = st.sm_boff; a.task_base3 = d_task_base;
    if (st.tot_sup) {
        if (j.fast) {
            const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
            EvPair ep{}; if (profile) { ep.a = ev_get(c); ep.b = ev_get(c); ep.kind = 4; ep.keys = st.tot_sup; (void)hipEventRecord(ep.a, c->stream); }
            if (bytes_mode) {
                ParseArgs ab = a;
                ab.place_group = std::max<u32>(1, std::min<u32>(PLACE_BYTES_TILES, PLACE_BYTES_REC / a.rec_cap));
                const size_t lds = (size_t)ntasks * 40 + PLACE_BYTES_REC * 8 + PLACE_BYTES_WORDS * 4 + (ab.sm_sub16 ? PLACE_BYTES_REC * 2 : 0);
                static bool announced = false;
                if (!announced) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(place_bytes_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); announced = true; }
                hipLaunchKernelGGL(place_bytes_kernel, dim3(j.nblocks), dim3(PLACE_BYTES_THREADS), lds, c->stream, ab);
            } else if (item_mode) {
                ParseArgs ai = a;
                ai.place_group = std::max<u32>(1, std::min<u32>(PLACE_ITEM_TILES, PLACE_ITEM_REC / a.rec_cap));
                hipLaunchKernelGGL(place_items_kernel, dim3(j.nblocks), dim3(PLACE_ITEM_THREADS), place_items_lds(ntasks), c->stream, ai);
            } else hipLaunchKernelGGL(place_kernel, dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16 + PLACE_MAX_REC * 4, c->stream, a);
            if (profile) { (void)hipEventRecord(ep.b, c->stream); c->ev_pending.push_back(ep); }
        }
        else if (a.dest_cache) hipLaunchKernelGGL(emit_kernel, dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        else if (a.drop_mask) hipLaunchKernelGGL((parse_kernel<PARSE_EMIT, false, true>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        else hipLaunchKernelGGL((parse_kernel<PARSE_EMIT, false>), dim3(j.nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, c->stream, a);
        if (ext) hipLaunchKernelGGL(resolve_pos_rid_kernel, dim3((u32)std::min<u64>((st.tot_sup + 255) / 256, 8192)), dim3(256), 0, c->stream,
                                    st.sm_gpos, st.tot_sup, j.d_roff, j.nreads, j.rid_base, st.sm_pos, st.sm_rid, (const u32 *)j.d_tile_r0, a.ntiles);
    }
    HIPCHK(c, hipGetLastError());
    // the small matrices are released after the stream has consumed them (pool reuse is stream-ordered:
    // every later user of these blocks is enqueued on the same stream)
    c->pool.release(d_blk_base); c->pool.release(d_task_tot); c->pool.release(d_task_base); c->pool.release(d_order); c->pool.release(d_skip);
    a.task_skip = nullptr;
    return HSK_OK;
}

// hsk_count() from pinned host memory on one GPU (no payload): ingest, scan and placement as ONE pipeline over slabs.
//   copy stream   DMA copy of slab s + 2          (the link: ~55 GB/s, the floor of the whole parse)
//   main stream   scan_kernel of slab s            (behind the copy of slab s + 1; the VALUs have ~0.8 ms to spare per slab)
//   second stream task totals, slot bases and place_kernel of slab s - 1   (fills those gaps: the placement, 8.6 ms when it
//                 runs after the last scan, is off the critical path)
// The store is laid out [slab][task] (a slab's slot bases need only the slabs before it: parse_scan_kernel carries the
// running totals on the device), so a task's supermers are one segment per slab (`segs`), which the extraction takes like the
// segments of several source ranks.  Returns HSK_OK, an error, or PARSE_FALLBACK: nothing has been decided, the packed reads
// are in HBM, the caller takes parse_count / parse_place (a tile beyond the record capacity, an index that needs the
// caller's offsets, a device-side index check that failed).
constexpr int PARSE_FALLBACK = -1000;
static int parse_ingest_pipelined(hsk_ctx *c, const u8 *h2d_src, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                                  int64_t rid_base, u32 ntasks, SupermerStore &st, std::vector<TaskSegs> &segs)
{
    u32 nblocks = 0;
    ParseArgs a = make_parse_args(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, ntasks, &nblocks, (u32)c->h2d_slabs);
    if (a.nslabs <= 1) return PARSE_FALLBACK;
    const u32 nsl = a.nslabs;
    const bool profile = (c->cfg.flags & HSK_FLAG_PROFILE) != 0;
    a.rec_cap = parse_rec_cap(c->cfg.kmer_size - c->cfg.minimizer_size + 1);
    a.place_group = std::max<u32>(1, std::min<u32>(16, PLACE_MAX_REC / a.rec_cap));
    a.place_one = 1;
    const bool bins = c->item_mode_now && c->combine_now && scan_place_enabled() && a.rec_cap <= PLACE_ITEM_REC;      // scan_kernel places the items itself: no records, no placement kernel
    ScanBins sbins;
    u64 *d_blk_cnt, *d_blk_base, *d_tot, *d_task_base, *d_run; u32 *d_order, *d_tile_rec = nullptr, *d_tile_nrec = nullptr, *d_overflow, *d_tile_sub = nullptr;
    const size_t mat = (size_t)nblocks * ntasks;
    DALLOC(c, d_blk_cnt, u64 *, mat * 3 * 8 * nsl);
    DALLOC(c, d_blk_base, u64 *, mat * 2 * 8);
    DALLOC(c, d_tot, u64 *, (size_t)nsl * ntasks * 3 * 8 + 64);
    DALLOC(c, d_task_base, u64 *, (size_t)ntasks * 3 * 8);
    DALLOC(c, d_run, u64 *, 256);
    DALLOC(c, d_order, u32 *, (size_t)ntasks * 4);
    if (!bins) { DALLOC(c, d_tile_rec, u32 *, (size_t)a.ntiles * a.rec_cap * 4 + 64); DALLOC(c, d_tile_nrec, u32 *, (size_t)a.ntiles * 4 + 64); }
    DALLOC(c, d_overflow, u32 *, 256);
    if (!bins && c->combine_now && a.rec_cap <= PLACE_ITEM_REC) DALLOC(c, d_tile_sub, u32 *, (size_t)a.ntiles * a.rec_cap * 4 + 64);
    if (bins) { int brc = bins_alloc(c, sbins, ntasks, packed_bytes, nreads, c->cfg.kmer_size - c->cfg.minimizer_size + 1, c->stream); if (brc) return brc; }
    u64 *d_ps; DALLOC(c, d_ps, u64 *, (size_t)PS_SEGS * ntasks * 3 * 8);
    unsigned long long *d_dropped = nullptr;                                // scan_kernel<.., DROP>: positions left out (hsk_ctx::drop_mask_now)
    // the store holds at most rec_cap supermers per tile (a tile beyond that falls back); its real size is known when the last slab is in
    const u64 cap_sup = a.ntiles * (u64)a.rec_cap;
    st = SupermerStore();
    st.ntasks = ntasks; st.nblocks = nblocks;
    if (bins) {}                                                            // (the items live in the bins' chunk store)
    else if (d_tile_sub) { DALLOC(c, st.sm_sub, u32 *, cap_sup * 4 + 64); DALLOC(c, st.sm_item, u64 *, cap_sup * 16 + 64); }      // item mode (combining extraction)
    else { DALLOC(c, st.sm_len, u8 *, cap_sup + 64); DALLOC(c, st.sm_gpos, u64 *, cap_sup * 8 + 64); }
    auto release_all = [&]() {
        bins_release(c, sbins);
        c->pool.release(d_tile_sub); c->pool.release(d_ps); c->pool.release(d_dropped);
        c->pool.release(d_blk_cnt); c->pool.release(d_blk_base); c->pool.release(d_tot); c->pool.release(d_task_base); c->pool.release(d_run);
        c->pool.release(d_order); c->pool.release(d_tile_rec); c->pool.release(d_tile_nrec); c->pool.release(d_overflow);
    };
    std::vector<u32> order(ntasks); for (u32 t = 0; t < ntasks; ++t) order[t] = t;
    st.order = order;
    u32 *h_order = (u32 *)((char *)c->pinned + (128u << 10));              // (pinned staging: the copy is asynchronous)
    memcpy(h_order, order.data(), (size_t)ntasks * 4);
    hipStream_t sA = c->stream, sB = c->comm_stream, sC = c->d2h_stream;
    HIPCHK(c, hipMemsetAsync(d_overflow, 0, 4, sA));
    HIPCHK(c, hipMemsetAsync(d_run, 0, 24, sA));
    HIPCHK(c, hipMemcpyAsync(d_order, h_order, (size_t)ntasks * 4, hipMemcpyHostToDevice, sA));
    a.tile_rec = d_tile_rec; a.tile_nrec = d_tile_nrec; a.overflow = d_overflow;
    a.sm_len = st.sm_len; a.sm_gpos = st.sm_gpos; a.blk_base = d_blk_base; a.task_base3 = d_task_base;
    a.tile_sub = d_tile_sub; a.sm_sub = st.sm_sub; a.sm_item = st.sm_item;
    if (d_tile_sub) a.place_group = std::max<u32>(1, std::min<u32>(PLACE_ITEM_TILES, PLACE_ITEM_REC / a.rec_cap));
    if (bins) bins_args(c, a, sbins);
    c->dropped_now = 0;
    a.drop_mask = bins ? 0u : c->drop_mask_now;
    if (a.drop_mask) { DALLOC(c, d_dropped, unsigned long long *, 256); HIPCHK(c, hipMemsetAsync(d_dropped, 0, 8, sA)); a.dropped = d_dropped; }
    EvList evs(c);
    hipEvent_t ready = evs.get();                                          // the small buffers above are set up; the second stream may start
    HIPCHK(c, hipEventRecord(ready, sA));
    HIPCHK(c, hipStreamWaitEvent(sB, ready, 0));
    // A slab's scan reaches PARSE_WORDS - 128 words into the NEXT slab (the windows of its last k-mers), not further: every slab's first
    // SLAB_HEAD bytes travel as a copy of their own with an event (`head`), and the scan of slab s waits for slab s and for the head of slab
    // s + 1 -- not for all of slab s + 1 (the first scan started after two slabs, 5.6 ms into the call; now after one)
    std::vector<hipEvent_t> landed(nsl), head(nsl), scanned(nsl);
    EvPair hp{}; if (profile) { hp.a = ev_get(c); hp.b = ev_get(c); hp.kind = 5; (void)hipEventRecord(hp.a, sC); }
    for (u32 sl = 0; sl < nsl; ++sl) {
        const u64 b0 = std::min<u64>((u64)sl * a.slab_tiles * (PARSE_TILE / 4), packed_bytes), b1 = std::min<u64>(((u64)sl + 1) * a.slab_tiles * (PARSE_TILE / 4), packed_bytes);
        const u64 bh = std::min<u64>(b0 + SLAB_HEAD, b1);
        if (bh > b0) HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed) + b0, h2d_src + b0, bh - b0, hipMemcpyHostToDevice, sC));
        head[sl] = evs.get();
        HIPCHK(c, hipEventRecord(head[sl], sC));
        if (b1 > bh) HIPCHK(c, hipMemcpyAsync(const_cast<u8 *>(d_packed) + bh, h2d_src + bh, b1 - bh, hipMemcpyHostToDevice, sC));
        landed[sl] = evs.get();
        HIPCHK(c, hipEventRecord(landed[sl], sC));
    }
    if (profile) { (void)hipEventRecord(hp.b, sC); c->ev_pending.push_back(hp); }
    EvPair sp{}, pp{};
    if (profile) { sp.a = ev_get(c); sp.b = ev_get(c); sp.kind = 3; sp.bytes = packed_bytes; (void)hipEventRecord(sp.a, sA);
                   pp.a = ev_get(c); pp.b = ev_get(c); pp.kind = 4; (void)hipEventRecord(pp.a, sB); }
    const bool scan_generic = tune("scan_generic", 0) != 0;
    for (u32 sl = 0; sl < nsl; ++sl) {
        HIPCHK(c, hipStreamWaitEvent(sA, landed[sl], 0));
        if (sl + 1 < nsl) HIPCHK(c, hipStreamWaitEvent(sA, head[sl + 1], 0));
        a.slab = sl; a.blk_cnt = d_blk_cnt + (size_t)sl * mat * 3;
        if (a.drop_mask) hipLaunchKernelGGL((scan_kernel<0, 0, false, true>), dim3(nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, sA, a);
        else if (a.bins) {
            if (a.k == 31 && a.m == 17 && !scan_generic) hipLaunchKernelGGL((scan_kernel<31, 17, true>), dim3(nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, sA, a);
            else hipLaunchKernelGGL((scan_kernel<0, 0, true>), dim3(nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, sA, a);
        }
        else if (a.k == 31 && a.m == 17 && !scan_generic) hipLaunchKernelGGL((scan_kernel<31, 17>), dim3(nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, sA, a);
        else if (a.k == 51 && a.m == 17 && !scan_generic) hipLaunchKernelGGL((scan_kernel<51, 17>), dim3(nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, sA, a);
        else hipLaunchKernelGGL((scan_kernel<0, 0>), dim3(nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16, sA, a);
        scanned[sl] = evs.get();
        HIPCHK(c, hipEventRecord(scanned[sl], sA));
        // the slab's placement on the second stream: totals, bases (behind everything the slabs before laid out), supermers to their slots
        HIPCHK(c, hipStreamWaitEvent(sB, scanned[sl], 0));
        { int prc = launch_parse_scan(c, sB, (const u64 *)a.blk_cnt, nblocks, ntasks, (const u32 *)d_order, (const u8 *)nullptr,
                                      d_tot + (size_t)sl * ntasks * 3, d_task_base, d_blk_base, d_run, d_ps); if (prc) return prc; }
        if (bins) {}                                                        // (placed by the scan)
        else if (d_tile_sub) hipLaunchKernelGGL(place_items_kernel, dim3(nblocks), dim3(PLACE_ITEM_THREADS), place_items_lds(ntasks), sB, a);
        else hipLaunchKernelGGL(place_kernel, dim3(nblocks), dim3(PARSE_THREADS), (size_t)ntasks * 16 + PLACE_MAX_REC * 4, sB, a);
    }
    if (profile) { (void)hipEventRecord(sp.b, sA); c->ev_pending.push_back(sp); (void)hipEventRecord(pp.b, sB); }
    // the verdicts and the totals
    u32 *h_flags = (u32 *)((char *)c->pinned + c->pinned_bytes - 320);
    std::vector<u64> tot((size_t)nsl * ntasks * 3);
    u64 *h_tot = (u64 *)((char *)c->pinned + (192u << 10));
    const bool staged = (size_t)nsl * ntasks * 24 <= (64u << 10);
    HIPCHK(c, hipMemcpyAsync(h_flags, d_overflow, 4, hipMemcpyDeviceToHost, sB));
    HIPCHK(c, hipMemcpyAsync(h_flags + 1, c->d_err, 4, hipMemcpyDeviceToHost, sB));
    if (bins) HIPCHK(c, hipMemcpyAsync(h_flags + 2, sbins.ctl, 4, hipMemcpyDeviceToHost, sB));
    if (d_dropped) HIPCHK(c, hipMemcpyAsync(h_flags + 4, d_dropped, 8, hipMemcpyDeviceToHost, sB));      // (sB is behind every scan: placed / scanned events)
    HIPCHK(c, hipMemcpyAsync(staged ? h_tot : tot.data(), d_tot, (size_t)nsl * ntasks * 24, hipMemcpyDeviceToHost, sB));
    hipEvent_t placed = evs.get();
    HIPCHK(c, hipEventRecord(placed, sB));
    HIPCHK(c, hipStreamWaitEvent(sA, placed, 0));                          // the extraction (main stream) reads the store
    HIPCHK(c, hsk_sync(c, sB));
    if (profile) { pp.keys = 0; c->ev_pending.push_back(pp); }
    if (staged) memcpy(tot.data(), h_tot, (size_t)nsl * ntasks * 24);
    bool fallback = *h_flags != 0;                                          // a tile with more supermers than the record capacity
    if (c->roff_check.valid() && !c->roff_check.get()) { fallback = true; c->roff_bad = true; }     // the reads do not lie back to back: the caller's offsets are needed
    if (c->index_unchecked && (h_flags[1] & 32u)) fallback = true;          // (parse_count reports it)
    if (bins && (h_flags[1] & (2u | 128u | 256u))) { fallback = true; (void)hipMemsetAsync(c->d_err, 0, 4, sA); }      // the chunk store or a bin's map ran out (run_pipeline: the call again, without virtual tasks)
    if (fallback) { HIPCHK(c, hsk_sync(c, sA)); release_all(); free_store(c, st); return PARSE_FALLBACK; }
    if (d_dropped) c->dropped_now = *(const unsigned long long *)(h_flags + 4);
    if (bins) {
        sbins.nchunks = std::min(h_flags[2], sbins.cap);
        int brc = bins_to_store(c, sbins, st, ntasks, c->vt_shift, sA); if (brc) { release_all(); return brc; }      // (main stream: behind the `placed` wait above)
    }
    c->index_unchecked = false;
    // segments: slab by slab, tasks in storage order inside a slab (what parse_scan_kernel laid out on the device)
    st.task_tot.assign((size_t)ntasks * 3, 0); st.task_base.assign((size_t)ntasks * 3, 0);
    segs.assign(ntasks, TaskSegs());
    u64 run_s = 0, run_b = 0;
    for (u32 sl = 0; sl < nsl; ++sl)
        for (u32 i = 0; i < ntasks; ++i) {
            const u32 t = order[i];
            const u64 *m = &tot[((size_t)sl * ntasks + t) * 3];
            if (m[0] && bins && !segs[t].segs.empty()) segs[t].segs[0].n_sup += m[0];      // (bins: one pseudo segment per task -- only its supermer total is used)
            else if (m[0]) {
                ExpSeg sg; sg.sup_off = run_s; sg.n_sup = m[0]; sg.byte_off = run_b; sg.kmer_off = segs[t].nkmers; sg.tile_start = 0;
                segs[t].segs.push_back(sg);
            }
            segs[t].nkmers += m[2];
            st.task_tot[3 * t] += m[0]; st.task_tot[3 * t + 1] += m[1]; st.task_tot[3 * t + 2] += m[2];
            run_s += m[0]; run_b += m[1];
        }
    st.tot_sup = run_s; st.tot_bytes = run_b; st.tot_kmers = 0;
    for (u32 t = 0; t < ntasks; ++t) st.tot_kmers += st.task_tot[3 * t + 2];
    if (!bins && run_s > cap_sup) { release_all(); free_store(c, st); return fail(c, HSK_ERR_INTERNAL, "supermer store overrun (%llu > %llu)", (unsigned long long)run_s, (unsigned long long)cap_sup); }
    if (profile && !c->ev_pending.empty()) c->ev_pending.back().keys = run_s;
    release_all();                                                          // (stream-ordered reuse: later users are enqueued behind the kernels above)
    return HSK_OK;
}

// count + place with a known storage order
static int parse_phase(hsk_ctx *c, const u8 *d_packed, u64 packed_bytes, const u64 *d_roff, const u32 *d_rlen, u64 nreads,
                       int64_t rid_base, u32 ntasks, const std::vector<u32> &order, SupermerStore &st)
{
    ParseJob j;
    int rc = parse_count(c, d_packed, packed_bytes, d_roff, d_rlen, nreads, rid_base, ntasks, j);
    if (rc == HSK_OK) rc = parse_place(c, j, order, st);
    parse_release(c, j);
    return rc;
}
