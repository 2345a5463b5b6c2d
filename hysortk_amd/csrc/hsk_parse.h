// hsk_parse.h -- minimizer -> task assignment and supermer emission (device).
//
// Replaces, for the reads of one rank:
//   FindKmerDestinationsParallel  reference src/kmerops.cpp:1010-1041  (a4)
//   SupermerEncoder::encode       reference src/kmerops.cpp:1109-1147  (a5)
//   ScatteredSupermers            reference include/kmerops.hpp:114     (a6, storage only)
//
// Design (not a translation): the packed DnaBuffer is treated as ONE base stream of
// 4*packed_bytes positions; a workgroup walks tiles of 2048 positions.  Per tile:
//   1. the tile's bytes (+K-1 bases of halo) are staged in LDS as byte-swapped words,
//   2. every position gets its canonical M-mer MurmurHash (rolling state is not needed: 64 bits
//      are pulled from LDS at any bit offset),
//   3. every k-mer position gets min(hash) over its K-M+1 m-mers (shared-middle trick: 8
//      consecutive positions of a lane share all but 14 window elements) and dest = min % ntasks,
//   4. supermer starts are local decisions: a k-mer starts a supermer iff it is the first k-mer of
//      its read, its dest differs from the previous k-mer, or its position is a multiple of 128.
// The 128-position cut replaces the reference's "250 bases counted from the run start" cap
// (kmerops.cpp:1120), which is a serial dependence along the read; cutting at fixed positions
// keeps every decision tile-local and bounds a supermer to 128 k-mers (len <= 127+K <= 222 < 256,
// one byte).  Supermer boundaries are an internal wire format (only this library reads them);
// the multiset of k-mers (and of (k-mer, pos, rid) with EXTENSION) per task is unchanged.
//
// Two launches of the same kernel: COUNT (per workgroup x task: supermers, bytes, k-mers) and,
// after an exclusive scan, EMIT with per-(workgroup, task) cursors in LDS: no global atomics, and
// the supermers of a task that go to one destination rank are contiguous for the all-to-all.
#pragma once
#include "hsk_device.h"

namespace hsk {

constexpr int PARSE_THREADS = 256;
constexpr int PARSE_PPT = 8;                          // positions per thread
constexpr int PARSE_TILE = PARSE_THREADS * PARSE_PPT; // 2048 positions = 512 bytes
constexpr int SUPERMER_CUT = 128;                     // forced supermer boundary period (positions)
constexpr int MAX_K = 95;
constexpr int HSK_MAX_TASKS = 1024;
constexpr int PARSE_WORDS = PARSE_TILE / 16 + 12;     // 128 tile words + halo (K-1 <= 94 bases = 6 words) + overread
constexpr int PARSE_HMAX = PARSE_TILE + 96;           // hashes for TILE + (K-M) positions
constexpr int PARSE_RWIN = 64;                        // reads looked at per tile by the wave-parallel index search
static_assert(PARSE_RWIN + PARSE_WORDS <= PARSE_THREADS, "scan_kernel prefetches the next tile with one lane per read-index entry and per word");

struct BinTable { u32 *cursor; u32 *map; u32 vmax; u32 cap_chunks; u32 *ctl; u32 *chunk_bin; ulonglong2 *items; u32 *subs; u32 *err; };

struct ParseArgs {
    const u8 *packed;          // 4-byte aligned; nothing beyond packed_bytes is read
    u64 packed_bytes;
    const u64 *roff;           // nreads + 1 byte offsets (last = packed_bytes)
    const u32 *rlen;           // nreads
    u64 nreads;
    int k, m;
    u32 ntasks;
    FastMod fm;
    u64 ntiles;
    u32 tiles_per_block;
    // Slabs (hsk_count() from pinned host memory: the packed reads arrive slab by slab over PCIe while the slabs before are
    // hashed): workgroup b owns, inside every slab s, the tiles [s * slab_tiles + b * tiles_per_block, + tiles_per_block).
    // nslabs <= 1: one range per workgroup as ever.  scan_kernel takes slab `slab` per launch and adds its counts to blk_cnt
    // (slab > 0); the placement kernels walk all slabs of the workgroup in the same order.
    u32 nslabs, slab;
    u64 slab_tiles;
    u32 place_one;             // placement kernels: only slab `slab` (its own blk_base: the slabs are placed one by one while the next ones are hashed)
    int64_t rid_base;
    // COUNT: blk_cnt[block][task][3] = {supermers, bytes, kmers}
    u64 *blk_cnt;
    // EMIT: blk_base[block][task][2] = {first supermer slot, first byte} (absolute)
    const u64 *blk_base;
    u8 *sm_len;                // supermer length in bases (one byte each)
    u64 *sm_gpos;              // position of the supermer's first base in the rank's packed base stream
    // DUMP (stage test): dest per base position, -1 where no k-mer starts
    int32_t *dump_dest;
    // optional: task id per base position (0xFFFF = no k-mer starts there), written by COUNT and read
    // by EMIT so that the minimizer hashes are computed once; 16-byte aligned, 2 B per position
    u16 *dest_cache;
    // fast path (scan_kernel / place_kernel): the supermers of a tile as compact records, kept from the
    // scan to the placement so that nothing is hashed or re-derived twice
    u32 *tile_rec;             // [ntiles][rec_cap]: position in tile | (k-mers - 1) << 11 | task << 18, in position order
    u32 *tile_nrec;            // [ntiles] supermers that start in the tile (may exceed rec_cap: see overflow)
    u32 rec_cap;
    u32 place_group;           // tiles placed together by one place_kernel step (rec_cap * place_group <= PLACE_MAX_REC)
    u32 *overflow;             // set when a tile holds more than rec_cap supermers (the host then takes parse_kernel)
    u32 *tile_r0;              // optional [ntiles] (EXTENSION): first read overlapping the tile, kept for resolve_pos_rid_kernel
    // optional (combining extraction, hsk_combine.h): 32 well-mixed bits of every supermer's minimizer hash, [ntiles][rec_cap] beside
    // tile_rec; the placement carries them to sm_sub[slot].  Supermers with the same minimizer m-mer -- all instances of a canonical
    // k-mer among them -- get the same value, independent of the task id (which is the hash modulo the task count)
    u32 *tile_sub;
    u32 *sm_sub;
    // scan_kernel places the items ITSELF (round 4; one GPU, combining extraction): no tile records, no placement kernel.  A bin = (XCD of the
    // workgroup, virtual task); a bin is a list of chunks of BIN_CHUNK items: cursor[bin] counts the items reserved, map[bin][v] names the
    // physical chunk of the bin's v-th chunk (+ 1; allocated from bin_ctl[0] by the lane whose item opens the chunk: the protocol of
    // hsk_scatter.h), chunk_bin[chunk] says whose it is.  All of a bin's writers run on ONE XCD: cursors, map words and the short runs that
    // fill a 128-byte line one 16-byte item at a time stay in that XCD's L2.
    u32 item_maxk;             // combining extraction: k-mers per item at most = min(16, 61 - K): the item is the supermer's first 64 bases, the last four of
                               // which give way to the k-mer count (0: 16)
    const struct BinTable *bins;   // the bins (device memory: scan_kernel<.., BINS = true> reads its fields where it needs them -- nine more pointers among the kernel's
                               // arguments cost the record-writing instances 12 % more instructions: the scalar registers spilled into vector lanes)
    unsigned short *sm_sub16;  // byte-store placement with several ranks (round 4): the top 16 of those bits beside sm_len -- they travel with the supermers, and the
                               // OWNER of a task builds the items and orders them by minimizer bucket (hsk_combine.h: items_build_kernel)
    u32 vt_shift;              // virtual tasks: `ntasks` = real tasks << vt_shift, task id = (hash mod real tasks) << vt_shift | top vt_shift minimizer bits (fm is the real count's)
    u64 *sm_item;              // place_items_kernel: two words per supermer -- its first 64 bases, left-aligned, the k-mer count (<= 16) in the low byte of the second
    const u8 *task_skip;       // optional [ntasks]: supermers of these tasks are not stored (heavy-hitter tasks travel as k-mer lists)
    // byte-store placement (place_bytes_kernel): the supermer's re-aligned bases (SupermerEncoder::copy_bits, reference
    // src/kmerops.cpp:1096-1107: (len + 3) / 4 bytes, tail bits zero) are written next to its length, task by task, so that the
    // extraction streams a task's own bytes instead of pulling every line of the packed reads into every XCD's L2
    u8 *sm_bytes;              // tasks in storage order, a task's supermers in slot order
    u32 *sm_boff;              // [slot] first byte of the supermer, relative to its task's first byte
    const u64 *task_base3;     // [ntasks][3] slot / byte / k-mer base of every task (parse_scan_kernel)
    u32 *packed_copy;          // optional (scan_kernel): `packed` is pinned HOST memory read in place over PCIe; every tile's words are
                               // also written here (HBM, packed_bytes + 64), so the ingest is fused into the one pass that hashes the reads
    u32 drop_mask;             // scan_kernel<.., DROP>: bit 0 / 1 = positions whose k-mer is an A or T / a C or G homopolymer hold no k-mer (its count inside the
                               // plan's sample alone exceeds U: it cannot be in the result, hsk_api.hip dispatch_pipeline)
    unsigned long long *dropped;   // ... and how many positions that were (they stay k-mers of the input)
};

enum ParseMode { PARSE_COUNT = 0, PARSE_EMIT = 1, PARSE_DUMP = 2 };

// last r in [lo, hi] with roff[r] <= b   (precondition: roff[lo] <= b)
__device__ __forceinline__ u64 find_read(const u64 *roff, u64 lo, u64 hi, u64 b)
{
    while (lo < hi) {
        u64 mid = lo + (hi - lo + 1) / 2;
        if (roff[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Supermer runs of a tile from the per-position task ids in s_dest (0xFFFF = no k-mer starts there).
// Lane t looks at positions t, t+256, ...: boundary(p) = invalid(p) | p % 128 == 0 | dest[p] != dest[p-1]
// goes into a bit mask with wave ballots; a run that starts at p ends at the next boundary, found with a bit
// scan over that mask (at most two words: runs never cross a multiple of 128).  runs[j] = k-mers in the
// supermer starting at position j*256 + t, 0 if none starts there.  No serial walk over LDS.
__device__ __forceinline__ void supermer_runs(const u16 *s_dest, u64 *s_bnd, u32 (&runs)[PARSE_PPT])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool start[PARSE_PPT];
#pragma unroll
    for (int j = 0; j < PARSE_PPT; ++j) {
        const u32 p = j * PARSE_THREADS + tid;
        const u16 d = s_dest[p];
        const u16 dp = p ? s_dest[p - 1] : (u16)0xFFFF;
        const bool bnd = (d == 0xFFFF) || ((p & (SUPERMER_CUT - 1)) == 0) || (d != dp);
        start[j] = bnd && d != 0xFFFF;
        const u64 m = __ballot(bnd);
        if (lane == 0) s_bnd[j * 4 + wave] = m;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PARSE_PPT; ++j) {
        runs[j] = 0;
        if (!start[j]) continue;
        const u32 p = j * PARSE_THREADS + tid;
        const u32 lim = (p | (SUPERMER_CUT - 1)) + 1;
        const u32 w = p >> 6;
        u64 m = ((p & 63) == 63) ? 0ULL : (s_bnd[w] & (~0ULL << ((p & 63) + 1)));
        u32 nb = lim;
        if (m) nb = w * 64 + (u32)__builtin_ctzll(m);
        else if ((w & 1) == 0) { const u64 m2 = s_bnd[w + 1]; if (m2) nb = (w + 1) * 64 + (u32)__builtin_ctzll(m2); }
        runs[j] = nb - p;
    }
}

// canonical M-mer hash at base position p of the staged tile, M > 32: Mmer<2> / Mmer<3> (reference include/supermer.hpp:23,
// GetRep :299, GetHash :308-313 = murmur over all 16 / 24 key bytes)
template <int MW>
__device__ __forceinline__ u64 wide_mmer_hash(const u32 *s_words, u32 p, int M)
{
    Mer<MW> fw;
#pragma unroll
    for (int x = 0; x < MW; ++x) fw.w[x] = bits64_be32(s_words, 2u * p + 64u * (u32)x);
    fw.w[MW - 1] &= ~0ULL << (64 * MW - 2 * M);
    const Mer<MW> c = canonical<MW>(fw, M);
    return murmur64_words<MW>(c.w);
}

// ParseArgs::drop_mask: which of a lane's eight positions hold a k-mer that is a homopolymer of a base in the mask (bit 0: A or T, bit 1: C or G).
// The 64 bases from the lane's first position hold the k-mers of all eight (K <= 57); a k-mer of one base b is the pattern b b b ... over its 2 K bits.
__device__ __forceinline__ u32 homopolymer_positions(const u32 *s_words, int tid, int K, u32 drop_mask)
{
    const u64 d0 = bits64_be32(s_words, 16u * (u32)tid), d1 = bits64_be32(s_words, 16u * (u32)tid + 64u);
    u32 dm = 0;
#pragma unroll
    for (int i = 0; i < PARSE_PPT; ++i) {
        const u64 hi = i ? ((d0 << (2 * i)) | (d1 >> (64 - 2 * i))) : d0, lo = d1 << (2 * i);
        const u32 b = (u32)(hi >> 62);
        const u64 pat = ((b & 1u) ? 0x5555555555555555ULL : 0ULL) | ((b & 2u) ? 0xAAAAAAAAAAAAAAAAULL : 0ULL);
        const bool same = K <= 32 ? ((hi ^ pat) >> (64 - 2 * K)) == 0 : (hi == pat && ((lo ^ pat) >> (128 - 2 * K)) == 0);
        if (same && ((drop_mask >> ((b == 1u || b == 2u) ? 1 : 0)) & 1u)) dm |= 1u << i;
    }
    return dm;
}

template <int MODE, bool EXT, bool DROP = false>      // (DROP: ParseArgs::drop_mask is honoured -- instances of their own, the others keep their registers)
__global__ __launch_bounds__(PARSE_THREADS, 4) void parse_kernel(ParseArgs a)
{
    __shared__ u32 s_words[PARSE_WORDS];
    __shared__ u64 s_hash[PARSE_HMAX];
    __shared__ __attribute__((aligned(16))) u16 s_dest[PARSE_TILE + 8];   // sizes of the statics are multiples of 16 B (dynamic LDS base stays aligned)
    __shared__ u64 s_rng[4];                 // r0, r1 (first / last read overlapping the tile), window base, fast-path flag
    __shared__ u64 s_roff[PARSE_RWIN];       // roff[rb + i]: the read index window of this tile (one coalesced load)
    __shared__ u32 s_rlen[PARSE_RWIN];
    __shared__ u64 s_bnd[PARSE_TILE / 64];    // bit p: a supermer boundary lies before position p (start of a run or a gap)
    __shared__ u32 s_scan[12];               // [0..8) block-scan scratch, [8] records in this tile (48 B keeps the dynamic base 16-B aligned)
    extern __shared__ __attribute__((aligned(16))) u64 s_cur[]; // 16 B per task: COUNT {supermers<<40|k-mers, bytes}; EMIT {slot cursor, u32 tile count, u32 tile prefix}

    u32 ndrop = 0;                           // (COUNT, ParseArgs::drop_mask) positions this lane has left out
    const int tid = threadIdx.x;
    const int K = a.k, M = a.m, W = K - M + 1;
    const u64 total_pos = a.packed_bytes * 4;

    if (MODE == PARSE_COUNT) {
        for (u32 i = tid; i < 2 * a.ntasks; i += PARSE_THREADS) s_cur[i] = 0;      // [task] = {supermers << 40 | k-mers, bytes}
    } else if (MODE == PARSE_EMIT) {
        // s_cur[task] = absolute slot cursor of this workgroup; then u32 tile counts and prefixes (see step 4)
        for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_cur[t] = a.blk_base[((u64)blockIdx.x * a.ntasks + t) * 2];
    }
    __syncthreads();

    const u64 tile0 = (u64)blockIdx.x * a.tiles_per_block;
    // The read index is searched ONCE per workgroup; afterwards every tile starts from where the
    // previous one ended (tiles of a workgroup are consecutive) with one 64-wide coalesced probe.
    if (tid == 0) s_rng[1] = (tile0 < a.ntiles) ? find_read(a.roff, 0, a.nreads - 1, (tile0 * PARSE_TILE) >> 2) : 0;
    __syncthreads();
    const u64 RINF = ~0ULL >> 2;
    for (u32 ti = 0; ti < a.tiles_per_block; ++ti) {
        const u64 tile = tile0 + ti;
        if (tile >= a.ntiles) break;
        const u64 gbase = tile * PARSE_TILE;            // first base position of the tile
        const bool cached_emit = (MODE == PARSE_EMIT) && a.dest_cache != nullptr;   // uniform: EMIT only replays the task ids
        const u64 bbase = gbase >> 2;                   // first byte

        // ---- 1. stage bytes (big-endian words) ------------------------------------------------
        if (!cached_emit) {
            const u32 *src = reinterpret_cast<const u32 *>(a.packed + bbase);   // bbase % 512 == 0, base 4-B aligned
            const u64 left = a.packed_bytes - bbase;                            // bytes readable from bbase
            for (int i = tid; i < PARSE_WORDS; i += PARSE_THREADS) {
                u32 wv = 0;
                if ((u64)i * 4 + 4 <= left) wv = __builtin_bswap32(src[i]);
                else if ((u64)i * 4 < left) {                                   // last partial word: byte loads only
                    for (u64 b = (u64)i * 4; b < left; ++b) wv |= (u32)a.packed[bbase + b] << (24 - 8 * (b & 3));
                }
                s_words[i] = wv;
            }
        }
        // tile-level read range (reads overlapping [bbase, bbase + 512)): wave 0 probes 64 index entries
        if (!cached_emit && tid < PARSE_RWIN) {
            const u64 rb = s_rng[1];                                   // <= r0 of this tile (offsets are monotone)
            u64 blast = bbase + PARSE_TILE / 4 - 1;
            if (blast >= a.packed_bytes) blast = a.packed_bytes - 1;
            const u64 idx = rb + tid;
            const u64 off = (idx <= a.nreads) ? a.roff[idx] : RINF;
            s_roff[tid] = off;
            s_rlen[tid] = (idx < a.nreads) ? a.rlen[idx] : 0;
            const u32 c0 = (u32)__popcll(__ballot(idx < a.nreads && off <= bbase));
            const u32 c1 = (u32)__popcll(__ballot(idx < a.nreads && off <= blast));
            if (tid == 0) {
                u64 r0 = rb + c0 - 1, r1 = rb + c1 - 1;
                const bool fast = c1 < PARSE_RWIN;                     // r1 + 1 is still inside the window
                if (!fast) {                                           // > 62 reads start in this tile (tiny reads): generic search
                    r0 = find_read(a.roff, rb, a.nreads - 1, bbase);
                    r1 = find_read(a.roff, r0, a.nreads - 1, blast);
                }
                s_rng[0] = r0; s_rng[1] = r1; s_rng[2] = rb; s_rng[3] = fast ? 1 : 0;
            }
        }
        __syncthreads();

        const int p0 = tid * PARSE_PPT;
        const bool cached = cached_emit;
        // ---- 2. canonical m-mer hashes for positions [0, TILE + W - 1) ---------------------------
        const u64 mmask = ~0ULL << (64 - 2 * (M < 32 ? M : 31));
        if (!cached) {
            if (M < 32) {
                for (int p = tid; p < PARSE_TILE + W - 1; p += PARSE_THREADS) {
                    u64 fw = bits64_be32(s_words, 2u * (u32)p) & mmask;
                    u64 tw = twin1(fw, M);
                    s_hash[p] = murmur64_8(tw < fw ? tw : fw);
                }
            } else if (M < 64) {
                for (int p = tid; p < PARSE_TILE + W - 1; p += PARSE_THREADS) s_hash[p] = wide_mmer_hash<2>(s_words, (u32)p, M);
            } else {
                for (int p = tid; p < PARSE_TILE + W - 1; p += PARSE_THREADS) s_hash[p] = wide_mmer_hash<3>(s_words, (u32)p, M);
            }
        }
        __syncthreads();

        // ---- 3. window minima, dest, validity -------------------------------------------------
        u64 mn[PARSE_PPT];
        if (cached) {
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) mn[i] = 0;
        } else if (W >= PARSE_PPT) {
            u64 c = ~0ULL;
            for (int j = PARSE_PPT - 1; j <= W - 1; ++j) { u64 v = s_hash[p0 + j]; c = v < c ? v : c; }
            u64 suf = ~0ULL;
            mn[PARSE_PPT - 1] = c;
#pragma unroll
            for (int i = PARSE_PPT - 2; i >= 0; --i) { u64 v = s_hash[p0 + i]; suf = v < suf ? v : suf; mn[i] = suf < c ? suf : c; }
            u64 run = ~0ULL;
#pragma unroll
            for (int i = 1; i < PARSE_PPT; ++i) { u64 v = s_hash[p0 + W + i - 1]; run = v < run ? v : run; mn[i] = run < mn[i] ? run : mn[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) {
                u64 c = ~0ULL;
                for (int j = 0; j < W; ++j) { u64 v = s_hash[p0 + i + j]; c = v < c ? v : c; }
                mn[i] = c;
            }
        }
        // read tracking for this lane's 8 positions (2 bytes)
        if (!cached) {
            const u64 g0 = gbase + p0;
            const u64 rlo = s_rng[0], rhi = s_rng[1], rb = s_rng[2];
            const bool fast = s_rng[3] != 0;
            u64 r, rstart, rend, nxt;
            if (fast) {                                                // everything this lane needs is in the LDS window
                u32 lo = (u32)(rlo - rb), hi = (u32)(rhi - rb);
                const u64 b0 = g0 >> 2;
                while (lo < hi) { u32 mid = (lo + hi + 1) >> 1; if (s_roff[mid] <= b0) lo = mid; else hi = mid - 1; }
                r = rb + lo; rstart = s_roff[lo] * 4; rend = rstart + s_rlen[lo];
                nxt = (r + 1 < a.nreads) ? s_roff[lo + 1] * 4 : ~0ULL;
            } else {
                r = find_read(a.roff, rlo, rhi, g0 >> 2);
                rstart = a.roff[r] * 4; rend = rstart + a.rlen[r];
                nxt = (r + 1 < a.nreads) ? a.roff[r + 1] * 4 : ~0ULL;
            }
            // tile-relative 32-bit positions from here on (64-bit compares per position are slow)
            const int HUGE = 1 << 30;
            auto rel = [&](u64 x) -> int { return x >= gbase + (u64)HUGE ? HUGE : (x < gbase ? ((gbase - x) >= (u64)HUGE ? -HUGE : -(int)(gbase - x)) : (int)(x - gbase)); };
            int rend_r = rel(rend), nxt_r = rel(nxt);
            const int total_r = rel(total_pos);
            u32 dmk = 0;                                               // (k-mers certain to be dropped: as scan_kernel<.., DROP>)
            if constexpr (DROP) dmk = homopolymer_positions(s_words, tid, K, a.drop_mask);
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) {
                const int g = p0 + i;
                while (g >= nxt_r) {
                    ++r; rstart = nxt;
                    if (fast) { const u32 j = (u32)(r - rb); rend = rstart + s_rlen[j]; nxt = (r + 1 < a.nreads) ? s_roff[j + 1] * 4 : ~0ULL; }
                    else { rend = rstart + a.rlen[r]; nxt = (r + 1 < a.nreads) ? a.roff[r + 1] * 4 : ~0ULL; }
                    rend_r = rel(rend); nxt_r = rel(nxt);
                }
                bool valid = (g < total_r) && (g + K <= rend_r);
                if (DROP && valid && ((dmk >> i) & 1u)) { valid = false; if (MODE == PARSE_COUNT) ++ndrop; }
                const u32 d = fastmod64(mn[i], a.fm);
                s_dest[p0 + i] = valid ? (u16)d : (u16)0xFFFF;
            }
        }
        if (cached) {                                   // 8 task ids = one 16-byte load per lane
            const uint4 v = *reinterpret_cast<const uint4 *>(a.dest_cache + gbase + p0);
            *reinterpret_cast<uint4 *>(&s_dest[p0]) = v;
        } else if (MODE == PARSE_COUNT && a.dest_cache != nullptr) {
            *reinterpret_cast<uint4 *>(a.dest_cache + gbase + p0) = *reinterpret_cast<const uint4 *>(&s_dest[p0]);
        }
        __syncthreads();

        // ---- 4. supermers ------------------------------------------------------------------------
        if (MODE == PARSE_DUMP) {
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) {
                u64 g = gbase + p0 + i;
                if (g < total_pos) { u16 d = s_dest[p0 + i]; a.dump_dest[g] = d == 0xFFFF ? -1 : (int32_t)d; }
            }
        } else if (MODE == PARSE_COUNT) {
            u32 runs[PARSE_PPT];
            supermer_runs(s_dest, s_bnd, runs);
#pragma unroll
            for (int j = 0; j < PARSE_PPT; ++j) {
                if (!runs[j]) continue;
                const u32 d = s_dest[j * PARSE_THREADS + tid];
                const u32 nk = runs[j];
                const u32 nb = (nk + K - 1 + 3) >> 2;
                atomicAdd((unsigned long long *)&s_cur[2 * d + 0], (1ULL << 40) | (unsigned long long)nk);   // supermers << 40 | k-mers
                atomicAdd((unsigned long long *)&s_cur[2 * d + 1], (unsigned long long)nb);
            }
        } else {
            // EMIT.  A supermer is the record {length, position of its first base}: the bases stay in the
            // packed reads (resident in HBM); bytes are materialised only for supermers that leave the GPU
            // (pack_kernel).  The tile's records are counting-sorted by task in LDS first, so that the global
            // stores of one wave instruction cover a few contiguous runs instead of 64 scattered slots.
            u32 *s_srt = reinterpret_cast<u32 *>(s_hash);                 // s_hash is dead here (>= 8 KB)
            u32 *s_tcnt = reinterpret_cast<u32 *>(s_cur + a.ntasks);       // [ntasks] records of this tile per task
            u32 *s_tpre = s_tcnt + a.ntasks;                               // [ntasks] exclusive prefix
            for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_tcnt[t] = 0;
            __syncthreads();
            u32 rec[PARSE_PPT], rnk[PARSE_PPT];
            {
                u32 runs[PARSE_PPT];
                supermer_runs(s_dest, s_bnd, runs);
#pragma unroll
                for (int j = 0; j < PARSE_PPT; ++j) {
                    rec[j] = 0xFFFFFFFFu;
                    if (!runs[j]) continue;
                    const u32 p = j * PARSE_THREADS + tid;
                    const u32 d = s_dest[p];
                    rec[j] = p | ((runs[j] - 1) << 11) | (d << 18);
                    rnk[j] = atomicAdd(&s_tcnt[d], 1u);
                }
            }
            __syncthreads();
            {   // exclusive prefix of the per-task counts: 4 consecutive tasks per lane
                u32 c4[4], sum = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const u32 t = tid * 4 + j; c4[j] = t < a.ntasks ? s_tcnt[t] : 0; sum += c4[j]; }
                u32 tot;
                u32 e = block_excl_scan_256<u32>(sum, s_scan, &tot);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const u32 t = tid * 4 + j; if (t < a.ntasks) s_tpre[t] = e; e += c4[j]; }
                if (tid == 0) s_scan[8] = tot;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i)
                if (rec[i] != 0xFFFFFFFFu) s_srt[s_tpre[rec[i] >> 18] + rnk[i]] = rec[i];
            __syncthreads();
            const u32 nrec = s_scan[8];
            for (u32 i = tid; i < nrec; i += PARSE_THREADS) {
                const u32 r = s_srt[i];
                const u32 d = r >> 18;
                if (a.task_skip && a.task_skip[d]) continue;
                const u64 slot = s_cur[d] + (i - s_tpre[d]);
                a.sm_len[slot] = (u8)(((r >> 11) & 127) + K);             // nk - 1 + K = bases in the supermer
                a.sm_gpos[slot] = gbase + (u64)(r & 2047);
            }
            __syncthreads();
            for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_cur[t] += s_tcnt[t];
        }
        __syncthreads();
    }

    if (MODE == PARSE_COUNT) {
        for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) {
            u64 *o = a.blk_cnt + ((u64)blockIdx.x * a.ntasks + t) * 3;
            const u64 pk = s_cur[2 * t];
            o[0] = pk >> 40; o[1] = s_cur[2 * t + 1]; o[2] = pk & ((1ULL << 40) - 1);
        }
        if (DROP && a.dropped) {
            u32 v = ndrop;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
            if (lane_id() == 0 && v) atomicAdd(a.dropped, (unsigned long long)v);
        }
    }
}

// EMIT when the task ids were kept by COUNT (dest_cache): nothing is hashed and no base is read, so this is a
// separate lean kernel (14 KB of LDS, few registers -> twice the residency of parse_kernel; the tile loop is a
// chain of short barrier-separated phases and lives on latency hiding).  Same tile -> workgroup mapping and
// the same per-(workgroup, task) cursors as parse_kernel<PARSE_EMIT>.
__global__ __launch_bounds__(PARSE_THREADS) void emit_kernel(ParseArgs a)
{
    __shared__ __attribute__((aligned(16))) u16 s_dest[PARSE_TILE + 8];
    __shared__ u32 s_srt[PARSE_TILE];
    __shared__ u64 s_bnd[PARSE_TILE / 64];
    __shared__ u32 s_scan[12];
    extern __shared__ __attribute__((aligned(16))) u64 s_cur[];      // [ntasks] slot cursors, then u32 tile counts, u32 tile prefixes
    const int tid = threadIdx.x;
    const int K = a.k;
    u32 *s_tcnt = reinterpret_cast<u32 *>(s_cur + a.ntasks);
    u32 *s_tpre = s_tcnt + a.ntasks;
    for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_cur[t] = a.blk_base[((u64)blockIdx.x * a.ntasks + t) * 2];
    __syncthreads();
    const u64 tile0 = (u64)blockIdx.x * a.tiles_per_block;
    uint4 nextv = make_uint4(0, 0, 0, 0);
    if (tile0 < a.ntiles) nextv = *reinterpret_cast<const uint4 *>(a.dest_cache + tile0 * PARSE_TILE + tid * PARSE_PPT);
    for (u32 ti = 0; ti < a.tiles_per_block; ++ti) {
        const u64 tile = tile0 + ti;
        if (tile >= a.ntiles) break;
        const u64 gbase = tile * PARSE_TILE;
        *reinterpret_cast<uint4 *>(&s_dest[tid * PARSE_PPT]) = nextv;
        if (ti + 1 < a.tiles_per_block && tile + 1 < a.ntiles)               // prefetch the next tile's task ids
            nextv = *reinterpret_cast<const uint4 *>(a.dest_cache + (gbase + PARSE_TILE) + tid * PARSE_PPT);
        for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_tcnt[t] = 0;
        __syncthreads();
        u32 rec[PARSE_PPT], rnk[PARSE_PPT];
        {
            u32 runs[PARSE_PPT];
            supermer_runs(s_dest, s_bnd, runs);
#pragma unroll
            for (int j = 0; j < PARSE_PPT; ++j) {
                rec[j] = 0xFFFFFFFFu;
                if (!runs[j]) continue;
                const u32 p = j * PARSE_THREADS + tid;
                const u32 d = s_dest[p];
                rec[j] = p | ((runs[j] - 1) << 11) | (d << 18);
                rnk[j] = atomicAdd(&s_tcnt[d], 1u);
            }
        }
        __syncthreads();
        {
            u32 c4[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const u32 t = tid * 4 + j; c4[j] = t < a.ntasks ? s_tcnt[t] : 0; sum += c4[j]; }
            u32 tot;
            u32 e = block_excl_scan_256<u32>(sum, s_scan, &tot);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const u32 t = tid * 4 + j; if (t < a.ntasks) s_tpre[t] = e; e += c4[j]; }
            if (tid == 0) s_scan[8] = tot;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PARSE_PPT; ++i)
            if (rec[i] != 0xFFFFFFFFu) s_srt[s_tpre[rec[i] >> 18] + rnk[i]] = rec[i];
        __syncthreads();
        const u32 nrec = s_scan[8];
        for (u32 i = tid; i < nrec; i += PARSE_THREADS) {
            const u32 r = s_srt[i];
            const u32 d = r >> 18;
            if (a.task_skip && a.task_skip[d]) continue;
            const u64 slot = s_cur[d] + (i - s_tpre[d]);
            a.sm_len[slot] = (u8)(((r >> 11) & 127) + K);
            a.sm_gpos[slot] = gbase + (u64)(r & 2047);
        }
        __syncthreads();
        for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_cur[t] += s_tcnt[t];
        __syncthreads();
    }
}

// ======================================================================================================
// Fast path: scan_kernel (minimizers, supermer records, per-(workgroup, task) counts) + place_kernel.
//
// Same contract as parse_kernel<COUNT> + emit_kernel, rebuilt around the instruction count (the parse is
// VALU-bound: ~10^10 positions x MurmurHash3; integer multiplies are full rate on gfx950, so every other
// instruction per position matters as much as the six 64-bit multiplies):
//   * a lane owns 8 CONSECUTIVE positions: one 64-bit window of the base stream gives all 8 m-mers by
//     constant shifts, the reverse strand is rolled (one base per step) instead of reversed per position;
//   * the lane keeps its 8 hashes in registers and reads only the W-1 neighbours' hashes from LDS;
//   * supermers are cut where the window MINIMUM changes (not where min % ntasks changes): no modulo per
//     position; a supermer is never longer than before, the k-mers of a task are the same multiset
//     (supermer boundaries are an internal format, see the header of this file);
//   * supermer starts are compacted first, then one lane per SUPERMER does the modulo, the run length
//     (bit scan over the boundary mask) and the per-task counters;
//   * the tile's supermers leave as 4-byte records in position order (tile_rec); place_kernel reads them
//     back (~1 KB per tile instead of the 4 KB task-id cache), counting-sorts a group of tiles by task in
//     LDS and writes {length, position} to the per-task slots.
// Requires 2*M + 14 <= 64 (M <= 25) so that the 8 m-mers of a lane fit one 64-bit window.
// ======================================================================================================
constexpr int SCAN_MAX_M = 25;
constexpr u32 BIN_CHUNK = 8192;                       // items per chunk of a scan-placed bin (= one staging step of bucket_scatter_kernel)
constexpr unsigned PARSE_XCC_GETREG = 20u | (0u << 6) | (3u << 11);      // HW_REG_XCC_ID, bits [3:0] (as hsk_sort.h)
constexpr u32 BIN_SPIN_LIMIT = 1u << 12;               // (~0.25 M cycles: see bin_slot)
constexpr u32 BIN_CUR_STRIDE = 32;                    // a bin's 32-bit cursor has a 128-byte line to itself: the L2 takes the atomics of one LINE one after the other, and 16 cursors
                                                      // to a line made 40 hot lines per XCD carry 131 M atomics each call (measured: the scan 75 instead of 60 ms)

// one item into its bin, in three steps that scan_kernel keeps a tile apart: (1) a slot is reserved with an atomic on the bin's cursor; (2) the
// chunk of that slot is found (or opened) through the map -- the workgroup remembers the last chunk of every virtual task in LDS (`mc`), so only
// one lookup in a few thousand goes to memory; (3) the item is stored.  Returns 0xFFFFFFFF (and sets error bit 128 / 256 / 2: chunk store or map
// exhausted, a chunk that never appeared -- the host runs the call again without the combining extraction) when there is no slot.
__device__ __forceinline__ u32 bin_reserve(const BinTable &a, u32 bin)
{
    return __hip_atomic_fetch_add(&a.cursor[(u64)bin * BIN_CUR_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ u64 bin_slot(const BinTable &a, u32 bin, u32 p, unsigned long long *mc /* LDS: {v << 32 | chunk + 1} of virtual task vt at mc[2 vt] */, u32 vt)
{
    typedef __attribute__((address_space(1))) u32 G32;
    u32 *const err = a.err;
    const u32 v = p / BIN_CHUNK, off = p % BIN_CHUNK;
    if (v >= a.vmax) { atomicOr(err, 256u); return ~0ULL; }
    // Chunks are opened AHEAD: a bin's first chunk by the host (bins_init_kernel), chunk v + 1 by the item that lands in the middle of chunk v.  Nobody
    // ever waits for a chunk whose opener has not even reserved its slot yet -- an item resolves its slot up to a tile after it reserved it, and a lane that
    // waited for an opener of its own workgroup's current tile would wait for ever (both sit behind the same barrier).  A bin would have to take half a
    // chunk of items within one tile's time to overrun the opener: only an input of ONE repeated minimizer does that, and it gets error bit 2 after a
    // short wait (the call then runs again without the combining extraction) instead of a slot.
    if (off == BIN_CHUNK / 2) {
        if (v + 1 >= a.vmax) atomicOr(err, 256u);
        else {
            u32 nx = __hip_atomic_fetch_add(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (nx > a.cap_chunks) { atomicOr(err, 128u); nx = 0xFFFFFFFFu; }
            else a.chunk_bin[nx - 1u] = bin | ((v + 1u) << 13);
            __hip_atomic_store((G32 *)(a.map + (u64)bin * a.vmax + v + 1u), nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    u32 ph;
    const unsigned long long e = mc[2u * vt];
    if ((u32)(e >> 32) == v && (u32)e != 0u) ph = (u32)e;
    else {
        G32 *mp = (G32 *)(a.map + (u64)bin * a.vmax + v);
        u32 spins = 0;
        while ((ph = __hip_atomic_load(mp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
            if (++spins > BIN_SPIN_LIMIT) { atomicOr(err, 2u); ph = 0xFFFFFFFFu; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        mc[2u * vt] = ((unsigned long long)v << 32) | ph;
    }
    if (ph == 0xFFFFFFFFu) return ~0ULL;                                 // (the chunk store ran out when this chunk was opened)
    return (u64)(ph - 1u) * BIN_CHUNK + off;
}
// a bin's first chunk: chunk `bin` (the allocator starts behind them)
__global__ void bins_init_kernel(BinTable t, u32 nbins)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbins) { t.map[(u64)b * t.vmax] = b + 1u; t.chunk_bin[b] = b; }
    if (b == 0) t.ctl[0] = nbins;
}
__device__ __forceinline__ void bin_store(const BinTable &a, u64 slot, u64 w0, u64 w1, u32 nk, u32 sub)
{
    if (slot == ~0ULL) return;
    a.items[slot] = make_ulonglong2(w0, (w1 & ~0xFFULL) | (u64)nk);
    a.subs[slot] = sub;
}

constexpr u32 SCAN_REC_CAP = 512;                     // default records kept per tile (expected ~270 at K=31, M=17)
constexpr u32 PLACE_MAX_REC = 8192;                   // records of one placement step (rec_cap * place_group)
// LDS layout of the tile's hashes: position p = 8*t + i lives at [i][t] (row stride SCAN_HSTRIDE), so that the 64
// lanes of a wave, which all touch the same i of consecutive t, hit consecutive 8-byte words (a lane-major layout
// puts them 64 bytes apart: 8-way bank conflicts on every access)
constexpr int SCAN_HSTRIDE = PARSE_THREADS + 16;      // 256 lanes + 12 lanes' worth of positions behind the tile
__device__ __forceinline__ int scan_hidx(int p) { return (p & 7) * SCAN_HSTRIDE + (p >> 3); }

#ifdef HSK_DIAG
__device__ unsigned long long g_scan_diag[16];
#endif
// KT, MT: k and m as compile-time constants (0: taken from the arguments).  The reference fixes both at compile time
// (KMER_SIZE, MINIMIZER_SIZE); here the default pair gets its own instance: shifts, masks and the window loop fold.
template <int KT, int MT, bool BINS = false, bool DROP = false>      // BINS: the items of the combining extraction are placed by this kernel (ParseArgs::bins) instead of records being written; DROP: ParseArgs::drop_mask (an instance of its own: the others keep their registers)
__global__ __launch_bounds__(PARSE_THREADS, 4) void scan_kernel(ParseArgs a)
{
    __shared__ u32 s_words2[BINS ? 2 : 1][PARSE_WORDS];                             // the tile's packed words (+ halo); two buffers: the items of a tile are stored while the NEXT tile is hashed (bins)
    __shared__ uint2 s_pend[BINS ? 2 * PARSE_THREADS : 1];                          // scan-placed items: {position | k-mers << 11 | virtual task << 16, minimizer bits} of the supermers whose slots are on their way
    __shared__ __attribute__((aligned(16))) u64 s_hash[8 * SCAN_HSTRIDE]; // hashes of the tile ([i][t] layout); later the minima of the supermer starts
    __shared__ u64 s_last[PARSE_THREADS];                                // window minimum of every lane's last position
    __shared__ u64 s_bmin[PARSE_THREADS];                                // minimum of every lane's eight hashes (windows of 16 k-mers and more)
    __shared__ __attribute__((aligned(8))) u8 s_v8[PARSE_THREADS];       // valid mask of every lane's 8 positions
    __shared__ __attribute__((aligned(8))) u8 s_bnd8[PARSE_THREADS + 16]; // boundary mask: bit p = a supermer cannot continue across p
    __shared__ u16 s_plist[PARSE_TILE];                                  // positions of the supermer starts, ascending
    __shared__ u64 s_rng[4];
    __shared__ u64 s_roff[PARSE_RWIN];
    __shared__ u32 s_rlen[PARSE_RWIN];
    __shared__ u32 s_scan[12];
    extern __shared__ __attribute__((aligned(16))) u64 s_cur[];          // [task] {supermers << 40 | k-mers, bytes}; behind it (scan-placed items) [task] the chunk the workgroup last saw the bin in

    const int tid = threadIdx.x;
    const int K = KT ? KT : a.k, M = MT ? MT : a.m, W = K - M + 1;
    const u64 mmask = ~0ULL << (64 - 2 * M);
    for (u32 i = tid; i < 2 * a.ntasks; i += PARSE_THREADS) s_cur[i] = 0;
    // (scan-placed items: nobody needs the supermers' byte totals -- a task's second word is the workgroup's memory of the bin's last chunk instead:
    //  mc[vt] = s_cur[2 vt + 1]; a third word per task would cost the fourth workgroup per CU)
    unsigned long long *s_mc = reinterpret_cast<unsigned long long *>(s_cur) + 1;
    u32 pend_pos0 = 0, pend_pos1 = 0, pend_mask = 0;                     // slots reserved for this lane's (up to two) supermers of the previous tile, not yet resolved
    const u64 tile0 = (u64)a.slab * a.slab_tiles + (u64)blockIdx.x * a.tiles_per_block;
    const u64 tile_end = (a.nslabs > 1 && ((u64)a.slab + 1) * a.slab_tiles < a.ntiles) ? ((u64)a.slab + 1) * a.slab_tiles : a.ntiles;   // this launch's tiles end here
    if (tid == 0) s_rng[1] = (tile0 < a.ntiles) ? find_read(a.roff, 0, a.nreads - 1, (tile0 * PARSE_TILE) >> 2) : 0;
    __syncthreads();
    const u64 RINF = ~0ULL >> 2;
    const int p0 = tid * PARSE_PPT;
    const u32 maxk = a.item_maxk ? a.item_maxk : 16u;
    const u32 xcc_nvt = BINS ? (__builtin_amdgcn_s_getreg(PARSE_XCC_GETREG) & 7u) * a.ntasks : 0u;      // first bin of this workgroup's XCD
    u32 pf0 = 0, pf1 = 0, pf2 = 0; bool pf_have = false;      // prefetched: lanes < PARSE_RWIN {read offset, length}, the next PARSE_WORDS lanes one tile word each
    u32 ndrop = 0;                                            // (DROP) positions this lane has left out

#ifdef HSK_DIAG
    unsigned long long dacc[6] = {0, 0, 0, 0, 0, 0};
#endif
    // the items whose slots were reserved while the previous tile was cut into supermers are resolved and stored now (their words: the other buffer)
    auto complete_pending = [&](const u32 *wprev) {
        const BinTable &bt = *a.bins;
        if (pend_mask & 1u) {
            const uint2 m = s_pend[tid]; const u32 pp = m.x & 2047u, vt = m.x >> 16;
            bin_store(bt, bin_slot(bt, xcc_nvt + vt, pend_pos0, s_mc, vt), bits64_be32(wprev, 2u * pp), bits64_be32(wprev, 2u * pp + 64u), (m.x >> 11) & 31u, m.y);
        }
        if (pend_mask & 2u) {
            const uint2 m = s_pend[PARSE_THREADS + tid]; const u32 pp = m.x & 2047u, vt = m.x >> 16;
            bin_store(bt, bin_slot(bt, xcc_nvt + vt, pend_pos1, s_mc, vt), bits64_be32(wprev, 2u * pp), bits64_be32(wprev, 2u * pp + 64u), (m.x >> 11) & 31u, m.y);
        }
        pend_mask = 0;
    };
    u32 wsel = 0;
    for (u32 ti = 0; ti < a.tiles_per_block; ++ti) {
        const u64 tile = tile0 + ti;
        if (tile >= tile_end) break;
        const u64 gbase = tile * PARSE_TILE;
        const u64 bbase = gbase >> 2;
        u32 *const s_words = s_words2[BINS ? wsel : 0u];
        const u32 *const s_wprev = s_words2[BINS ? (wsel ^ 1u) : 0u];
        if (BINS) wsel ^= 1u;
#ifdef HSK_DIAG
        unsigned long long sd[6];
        if (tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); sd[0] = t_; }
#endif

        // ---- 1. stage bytes (big-endian words), read index window (as parse_kernel) -----------------------
        if (pf_have) { if (tid >= PARSE_RWIN && tid < PARSE_RWIN + PARSE_WORDS) s_words[tid - PARSE_RWIN] = __builtin_bswap32(pf0); }
        else {
            const u32 *src = reinterpret_cast<const u32 *>(a.packed + bbase);
            const u64 left = a.packed_bytes - bbase;
            for (int i = tid; i < PARSE_WORDS; i += PARSE_THREADS) {
                u32 wv = 0;
                if ((u64)i * 4 + 4 <= left) wv = __builtin_bswap32(src[i]);
                else if ((u64)i * 4 < left) {
                    for (u64 b = (u64)i * 4; b < left; ++b) wv |= (u32)a.packed[bbase + b] << (24 - 8 * (b & 3));
                }
                s_words[i] = wv;
            }
        }
        if (tid < PARSE_RWIN) {
            const u64 rb = s_rng[1];
            u64 blast = bbase + PARSE_TILE / 4 - 1;
            if (blast >= a.packed_bytes) blast = a.packed_bytes - 1;
            const u64 idx = rb + tid;
            const u64 off = pf_have ? ((u64)pf0 | ((u64)pf1 << 32)) : ((idx <= a.nreads) ? a.roff[idx] : RINF);
            s_roff[tid] = off;
            s_rlen[tid] = pf_have ? pf2 : ((idx < a.nreads) ? a.rlen[idx] : 0);
            const u32 c0 = (u32)__popcll(__ballot(idx < a.nreads && off <= bbase));
            const u32 c1 = (u32)__popcll(__ballot(idx < a.nreads && off <= blast));
            if (tid == 0) {
                u64 r0 = rb + c0 - 1, r1 = rb + c1 - 1;
                const bool fast = c1 < PARSE_RWIN;
                if (!fast) {
                    r0 = find_read(a.roff, rb, a.nreads - 1, bbase);
                    r1 = find_read(a.roff, r0, a.nreads - 1, blast);
                }
                s_rng[0] = r0; s_rng[1] = r1; s_rng[2] = rb; s_rng[3] = fast ? 1 : 0;
                if (a.tile_r0) a.tile_r0[tile] = (u32)r0;
            }
        }
        lds_barrier();
        if (a.packed_copy && tid < PARSE_TILE / 16 && bbase + 4u * (u64)tid < a.packed_bytes)
            a.packed_copy[(bbase >> 2) + tid] = __builtin_bswap32(s_words[tid]);      // (a last partial word is padded with zeros: the copy has room)
        // the next tile's word and read-index entries are requested now and land in LDS at the top of the next round (every
        // barrier of the loop is an LDS-only barrier: none of them waits for these loads or for the record stores)
        {
            const u64 nb = bbase + PARSE_TILE / 4;
            pf_have = (ti + 1 < a.tiles_per_block) && (tile + 1 < tile_end) && (nb + (u64)PARSE_WORDS * 4 <= a.packed_bytes);
            if (pf_have) {
                if (tid < PARSE_RWIN) {
                    const u64 idx = s_rng[1] + tid;                   // the read holding this tile's last byte is the next tile's first
                    const u64 o_ = (idx <= a.nreads) ? a.roff[idx] : RINF;
                    pf0 = (u32)o_; pf1 = (u32)(o_ >> 32);
                    pf2 = (idx < a.nreads) ? a.rlen[idx] : 0;
                } else if (tid < PARSE_RWIN + PARSE_WORDS) pf0 = reinterpret_cast<const u32 *>(a.packed + nb)[tid - PARSE_RWIN];
            }
        }
#ifdef HSK_DIAG
        if (tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); sd[1] = t_; }
#endif

        // ---- 2. canonical m-mer hashes of this lane's 8 positions (rolled), kept in registers -------------
        u64 h[PARSE_PPT];
        {
            const u64 w0 = bits64_be32(s_words, 16u * (u32)tid);       // 32 bases from position p0
            // reverse complement of the whole window: the twin of the m-mer at offset i is its bases [32 - M - i, 32 - i)
            // (i + M <= 32: SCAN_MAX_M), one shift per position instead of a rolled state
            const u64 rw = ~rev2(w0);
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) {
                const u64 fw = (w0 << (2 * i)) & mmask;
                const u64 rc = (rw << (2 * (32 - M - i))) & mmask;
                h[i] = murmur64_8(rc < fw ? rc : fw);
            }
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) s_hash[i * SCAN_HSTRIDE + tid] = h[i];
            if (W >= 2 * PARSE_PPT) {                                  // wide windows: the minimum of every lane's eight positions (step 3 takes whole lanes from these)
                u64 bm = h[0];
#pragma unroll
                for (int i = 1; i < PARSE_PPT; ++i) bm = h[i] < bm ? h[i] : bm;
                s_bmin[tid] = bm;
            }
            // the W-1 positions behind the tile (windows of the last k-mers): high lanes first
            for (int e = PARSE_THREADS - 1 - tid; e < W - 1; e += PARSE_THREADS) {
                const int p = PARSE_TILE + e;
                const u64 fw = bits64_be32(s_words, 2u * (u32)p) & mmask;
                const u64 tw = twin1(fw, M);
                s_hash[scan_hidx(p)] = murmur64_8(tw < fw ? tw : fw);
            }
        }
        lds_barrier();
#ifdef HSK_DIAG
        if (tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); sd[2] = t_; }
#endif

        // ---- 3. window minima (shared middle), validity --------------------------------------------------
        u64 mn[PARSE_PPT];
        if (W >= PARSE_PPT) {
            u64 c = h[PARSE_PPT - 1];
            // the shared middle [p0 + 8, p0 + W - 1]: whole lanes from their block minima (K = 51: 3 + 3 reads instead of 27), the rest position by position
            const int nfull = (W - PARSE_PPT) / PARSE_PPT;
            int j = PARSE_PPT;
            if (nfull > 0 && tid + nfull < PARSE_THREADS) {
                for (int b = 1; b <= nfull; ++b) { const u64 v = s_bmin[tid + b]; c = v < c ? v : c; }
                j = PARSE_PPT * (nfull + 1);
            }
#pragma unroll 8
            for (; j <= W - 1; ++j) { const u64 v = s_hash[scan_hidx(p0 + j)]; c = v < c ? v : c; }
            mn[PARSE_PPT - 1] = c;
            u64 suf = ~0ULL;
#pragma unroll
            for (int i = PARSE_PPT - 2; i >= 0; --i) { suf = h[i] < suf ? h[i] : suf; mn[i] = suf < c ? suf : c; }
            u64 run = ~0ULL;
#pragma unroll
            for (int i = 1; i < PARSE_PPT; ++i) { const u64 v = s_hash[scan_hidx(p0 + W + i - 1)]; run = v < run ? v : run; mn[i] = run < mn[i] ? run : mn[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) {
                u64 c = ~0ULL;
                for (int j = 0; j < W; ++j) { const u64 v = s_hash[scan_hidx(p0 + i + j)]; c = v < c ? v : c; }
                mn[i] = c;
            }
        }
        u32 vmask = 0;
        {
            const u64 g0 = gbase + p0;
            const u64 rlo = s_rng[0], rhi = s_rng[1], rb = s_rng[2];
            const bool fast = s_rng[3] != 0;
            u64 r, rstart, rend, nxt;
            if (fast) {
                u32 lo = (u32)(rlo - rb), hi = (u32)(rhi - rb);
                const u64 b0 = g0 >> 2;
                while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_roff[mid] <= b0) lo = mid; else hi = mid - 1; }
                r = rb + lo; rstart = s_roff[lo] * 4; rend = rstart + s_rlen[lo];
                nxt = (r + 1 < a.nreads) ? s_roff[lo + 1] * 4 : ~0ULL;
            } else {
                r = find_read(a.roff, rlo, rhi, g0 >> 2);
                rstart = a.roff[r] * 4; rend = rstart + a.rlen[r];
                nxt = (r + 1 < a.nreads) ? a.roff[r + 1] * 4 : ~0ULL;
            }
            if (nxt >= g0 + PARSE_PPT) {
                // one read under all 8 positions: k-mers start at g0 .. rend - K
                const long long cnt = (long long)rend - (long long)K - (long long)g0 + 1;
                vmask = cnt <= 0 ? 0u : (cnt >= PARSE_PPT ? 0xFFu : ((1u << (u32)cnt) - 1u));
            } else {
                // a read ends under this lane.  Reads start on bytes (4 positions) and g0 is a multiple of 8: the next read starts
                // at g0 + 4 exactly, positions 0-3 belong to read r and positions 4-7 to the LAST read that starts at g0 + 4 (reads
                // without bases share their offset with their successor); nothing else can start before g0 + 8.
                static_assert(PARSE_PPT == 8, "two bytes of bases per lane");
                const long long cnt_lo = (long long)rend - (long long)K - (long long)g0 + 1;
                vmask = cnt_lo <= 0 ? 0u : (cnt_lo >= 4 ? 0xFu : ((1u << (u32)cnt_lo) - 1u));
                do {
                    ++r; rstart = nxt;
                    if (fast) { const u32 j = (u32)(r - rb); rend = rstart + s_rlen[j]; nxt = (r + 1 < a.nreads) ? s_roff[j + 1] * 4 : ~0ULL; }
                    else { rend = rstart + a.rlen[r]; nxt = (r + 1 < a.nreads) ? a.roff[r + 1] * 4 : ~0ULL; }
                } while (nxt <= g0 + 4);
                const long long cnt_hi = (long long)rend - (long long)K - (long long)(g0 + 4) + 1;
                vmask |= (cnt_hi <= 0 ? 0u : (cnt_hi >= 4 ? 0xFu : ((1u << (u32)cnt_hi) - 1u))) << 4;
            }
        }
        if constexpr (DROP) {
            // positions whose k-mer is a homopolymer of a base in drop_mask: the 64 bases from this lane's first position hold the k-mers of all
            // eight (K <= 57); a k-mer of one base b is the pattern b b b ... over its 2 K bits
            const u32 dm = homopolymer_positions(s_words, tid, K, a.drop_mask);
            ndrop += (u32)__popc(vmask & dm);
            vmask &= ~dm;
        }
        s_last[tid] = mn[PARSE_PPT - 1];
        s_v8[tid] = (u8)vmask;
        lds_barrier();
#ifdef HSK_DIAG
        if (tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); sd[3] = t_; }
#endif

        // ---- 4. boundaries, supermer starts, compaction ---------------------------------------------------
        u32 start8 = 0;
        {
            // a supermer cannot continue across position i if i holds no k-mer, is a forced cut, follows a position without
            // k-mer or has another window minimum than its predecessor: all but the last as byte masks of the lane's 8 positions
            u64 pm = tid ? s_last[tid - 1] : 0;
            const u32 pv = tid ? (u32)(s_v8[tid - 1] >> 7) : 0u;
            u32 neq8 = 0;
#pragma unroll
            for (int i = 0; i < PARSE_PPT; ++i) { neq8 |= (mn[i] != pm ? 1u : 0u) << i; pm = mn[i]; }
            const u32 cut8 = ((tid & (SUPERMER_CUT / PARSE_PPT - 1)) == 0) ? 1u : 0u;
            const u32 prev8 = (vmask << 1) | pv;                         // bit i: position i - 1 holds a k-mer
            const u32 bnd8 = (~vmask | cut8 | ~prev8 | neq8) & 0xFFu;
            s_bnd8[tid] = (u8)bnd8;
            if (tid == 0) s_scan[10] = 0;                                // extra records of this tile (combining extraction, step 5)
            start8 = vmask & bnd8;
        }
        u32 nrec;
        u32 off = block_excl_scan_256_lds<u32>((u32)__popc(start8), s_scan, &nrec);   // (barriers inside: s_hash is free from here on)
#pragma unroll
        for (int i = 0; i < PARSE_PPT; ++i)
            if ((start8 >> i) & 1) { s_hash[off] = mn[i]; s_plist[off] = (u16)(p0 + i); ++off; }
        lds_barrier();
#ifdef HSK_DIAG
        if (tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); sd[4] = t_; }
#endif

        if (BINS) complete_pending(s_wprev);                      // (the previous tile's items: their slots have had a tile's time to arrive)
        // ---- 5. one lane per supermer: task, run length, counters, record ----------------------------------
        {
            const u64 *bw = reinterpret_cast<const u64 *>(s_bnd8);
            u32 *trec = a.tile_rec + tile * (u64)a.rec_cap;
            for (u32 r = tid; r < nrec; r += PARSE_THREADS) {
                const u32 p = s_plist[r];
                u32 d = fastmod64(s_hash[r], a.fm);
                const u32 sub = (u32)((s_hash[r] * 0x9E3779B97F4A7C15ULL) >> 32);
                if (a.vt_shift) d = (d << a.vt_shift) | (sub >> (32u - a.vt_shift));
                const u32 w = p >> 6;
                const u64 m = ((p & 63) == 63) ? 0ULL : (bw[w] & (~0ULL << ((p & 63) + 1)));
                u32 nb = (p | (SUPERMER_CUT - 1)) + 1;
                if (m) nb = w * 64 + (u32)__builtin_ctzll(m);
                else if ((w & 1) == 0) { const u64 m2 = bw[w + 1]; if (m2) nb = (w + 1) * 64 + (u32)__builtin_ctzll(m2); }
                u32 nk = nb - p;
                if constexpr (BINS) {
                    const BinTable &bt = *a.bins;
                    // the combining extraction on one GPU, items placed by this kernel.  The supermer's slot is RESERVED here (one atomic on its bin's
                    // cursor) and the answer is taken up a tile later (complete_pending, before the next tile's supermers are cut): the round trip
                    // to the L2 hides behind the next tile's hashes.  Only a lane's first two supermers of a tile travel that way; the rest -- more
                    // than 512 supermers in a tile, the pieces behind the first of a supermer longer than an item -- resolve at once.
                    const u32 bin = xcc_nvt + d;
                    for (u32 pp = p, rest = nk, first = 1u; rest; first = 0u) {
                        const u32 piece = rest < maxk ? rest : maxk;
                        atomicAdd((unsigned long long *)&s_cur[2 * d + 0], (1ULL << 40) | (unsigned long long)piece);
                        const u32 pos = bin_reserve(bt, bin);
                        if (first && r < 2u * PARSE_THREADS) {
                            s_pend[r] = make_uint2(pp | (piece << 11) | (d << 16), sub);
                            if (r < (u32)PARSE_THREADS) { pend_pos0 = pos; pend_mask |= 1u; } else { pend_pos1 = pos; pend_mask |= 2u; }
                        } else bin_store(bt, bin_slot(bt, bin, pos, s_mc, d), bits64_be32(s_words, 2u * pp), bits64_be32(s_words, 2u * pp + 64u), piece, sub);
                        pp += piece; rest -= piece;
                    }
                    continue;
                }
                if (a.tile_sub && nk > maxk) {
                    // combining extraction (hsk_combine.h): no supermer longer than 16 k-mers -- one work item, one 16-byte record.  A window
                    // minimum lives for at most W <= 15 positions unless its m-mer repeats (homopolymers, tandem repeats): rare; the rest of
                    // such a run leaves as extra records of 16 k-mers behind the tile's regular ones (their order does not matter)
                    for (u32 pp = p + maxk, rest = nk - maxk; rest; ) {
                        const u32 piece = rest < maxk ? rest : maxk;
                        const u32 rx = nrec + atomicAdd(&s_scan[10], 1u);
                        atomicAdd((unsigned long long *)&s_cur[2 * d + 0], (1ULL << 40) | (unsigned long long)piece);
                        atomicAdd((unsigned long long *)&s_cur[2 * d + 1], (unsigned long long)((piece + K - 1 + 3) >> 2));
                        if (rx < a.rec_cap) { trec[rx] = pp | ((piece - 1) << 11) | (d << 18); a.tile_sub[tile * (u64)a.rec_cap + rx] = sub; }
                        pp += piece; rest -= piece;
                    }
                    nk = maxk;
                }
                atomicAdd((unsigned long long *)&s_cur[2 * d + 0], (1ULL << 40) | (unsigned long long)nk);
                atomicAdd((unsigned long long *)&s_cur[2 * d + 1], (unsigned long long)((nk + K - 1 + 3) >> 2));
                if (r < a.rec_cap) { trec[r] = p | ((nk - 1) << 11) | (d << 18); if (a.tile_sub) a.tile_sub[tile * (u64)a.rec_cap + r] = sub; }
            }
        }
        lds_barrier();
        if (tid == 0 && !BINS) {
            const u32 nall = nrec + s_scan[10];                           // (the extra records have been counted: barrier above)
            a.tile_nrec[tile] = nall;
            if (nall > a.rec_cap) atomicOr(a.overflow, 1u);
        }
#ifdef HSK_DIAG
        if (tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); sd[5] = t_; }
#endif
#ifdef HSK_DIAG
        if (tid == 0) { for (int q = 0; q < 5; ++q) dacc[q] += sd[q + 1] - sd[q]; dacc[5] += 1; }
#endif
    }
#ifdef HSK_DIAG
    if (tid == 0) { for (int q = 0; q < 5; ++q) atomicAdd(&g_scan_diag[q], dacc[q]); atomicAdd(&g_scan_diag[8], dacc[5]); }
#endif
    if (BINS) complete_pending(s_words2[BINS ? (wsel ^ 1u) : 0u]);              // the last tile's items (its words: the buffer the loop used last)
    if constexpr (DROP) {
        u32 v = ndrop;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
        if (lane_id() == 0 && v) atomicAdd(a.dropped, (unsigned long long)v);
    }
    for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) {
        u64 *o = a.blk_cnt + ((u64)blockIdx.x * a.ntasks + t) * 3;
        const u64 pk = s_cur[2 * t];
        const u64 bytes_t = BINS ? 0ULL : s_cur[2 * t + 1];
        if (a.slab == 0 || a.place_one) { o[0] = pk >> 40; o[1] = bytes_t; o[2] = pk & ((1ULL << 40) - 1); }      // (place_one: every slab has its own matrix)
        else { o[0] += pk >> 40; o[1] += bytes_t; o[2] += pk & ((1ULL << 40) - 1); }      // (same workgroup, launches in stream order)
    }
}

// Placement: the records of `place_group` consecutive tiles of this workgroup are loaded (coalesced), counting-sorted
// by task in LDS and written to the per-(workgroup, task) slot ranges that parse_scan_kernel laid out.  Several
// tiles per step make the runs per task longer (~27 records of 8 + 1 bytes at 40 tasks) than one tile would.
// dynamic LDS: u64 cur[ntasks], u32 tcnt[ntasks], u32 tpre[ntasks], u32 srt[PLACE_MAX_REC]
__global__ __launch_bounds__(PARSE_THREADS) void place_kernel(ParseArgs a)
{
    // pipelined ingest (place_one): the placement of a slab is launched before the host has seen the scan's verdict.  A tile beyond the record
    // capacity means that the supermers counted (all of them) may exceed the store (rec_cap per tile): nothing is placed, the host falls back.
    if (a.place_one && a.overflow && *(const volatile u32 *)a.overflow) return;
    constexpr int RPT = PLACE_MAX_REC / PARSE_THREADS;                 // records per thread and step at most
    __shared__ u32 s_scan[12];
    __shared__ u32 s_go[20];                                            // record offsets of the tiles of the step (+ total)
    extern __shared__ __attribute__((aligned(16))) u64 s_cur[];
    const int tid = threadIdx.x;
    const int K = a.k;
    u32 *s_tcnt = reinterpret_cast<u32 *>(s_cur + a.ntasks);
    u32 *s_tpre = s_tcnt + a.ntasks;
    u32 *s_srt = s_tpre + a.ntasks;
    for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_cur[t] = a.blk_base[((u64)blockIdx.x * a.ntasks + t) * 2];
    const u32 G = a.place_group;
    const u32 nsl = a.nslabs > 1 ? a.nslabs : 1;
    for (u32 sl = a.place_one ? a.slab : 0; sl < (a.place_one ? a.slab + 1 : nsl); ++sl) {
    const u64 tile0 = (u64)sl * a.slab_tiles + (u64)blockIdx.x * a.tiles_per_block;
    const u64 tile_end = (nsl > 1 && ((u64)sl + 1) * a.slab_tiles < a.ntiles) ? ((u64)sl + 1) * a.slab_tiles : a.ntiles;
    for (u32 t0 = 0; t0 < a.tiles_per_block; t0 += G) {
        const u64 tfirst = tile0 + t0;
        if (tfirst >= tile_end) break;
        u32 ng = a.tiles_per_block - t0; if (ng > G) ng = G;
        if (tfirst + ng > tile_end) ng = (u32)(tile_end - tfirst);
        __syncthreads();                                                // previous step done with s_go / s_tcnt / s_srt
        if (tid == 0) {
            u32 run = 0;
            for (u32 j = 0; j < ng; ++j) { s_go[j] = run; u32 n = a.tile_nrec[tfirst + j]; run += n < a.rec_cap ? n : a.rec_cap; }
            for (u32 j = ng; j <= 16; ++j) s_go[j] = run;
        }
        for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_tcnt[t] = 0;
        __syncthreads();
        const u32 total = s_go[16];
        u32 rec[RPT], rnk[RPT];
#pragma unroll
        for (int x = 0; x < RPT; ++x) {
            const u32 i = x * PARSE_THREADS + tid;
            rec[x] = 0xFFFFFFFFu;
            if (i < total) {
                u32 j = 0;
                while (j + 1 < ng && s_go[j + 1] <= i) ++j;
                const u32 r = a.tile_rec[(tfirst + j) * (u64)a.rec_cap + (i - s_go[j])];
                rec[x] = r | (j << 28);
                rnk[x] = atomicAdd(&s_tcnt[(r >> 18) & 1023u], 1u);
            }
        }
        __syncthreads();
        {
            u32 c4[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const u32 t = tid * 4 + j; c4[j] = t < a.ntasks ? s_tcnt[t] : 0; sum += c4[j]; }
            u32 e = block_excl_scan_256<u32>(sum, s_scan, nullptr);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const u32 t = tid * 4 + j; if (t < a.ntasks) s_tpre[t] = e; e += c4[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < RPT; ++x)
            if (rec[x] != 0xFFFFFFFFu) s_srt[s_tpre[(rec[x] >> 18) & 1023u] + rnk[x]] = rec[x];
        __syncthreads();
        for (u32 i = tid; i < total; i += PARSE_THREADS) {
            const u32 r = s_srt[i];
            const u32 d = (r >> 18) & 1023u;
            if (a.task_skip && a.task_skip[d]) continue;
            const u64 slot = s_cur[d] + (i - s_tpre[d]);
            a.sm_len[slot] = (u8)(((r >> 11) & 127) + K);
            a.sm_gpos[slot] = (tfirst + (r >> 28)) * PARSE_TILE + (u64)(r & 2047);
        }
        __syncthreads();
        for (u32 t = tid; t < a.ntasks; t += PARSE_THREADS) s_cur[t] += s_tcnt[t];
    }
    }                                                                   // slabs
}

// Item placement (combining extraction, hsk_combine.h): same job as place_kernel, but what lands in a supermer's slot is the supermer
// ITSELF -- two words: its first 64 bases left-aligned (a supermer of <= 16 k-mers has at most K + 15 <= 47 of them for one-word keys),
// the k-mer count in the low byte of the second word -- and the 32 minimizer bits beside it (sm_sub).  The bases are read here, once
// and in order, from the packed words of the step's tiles staged in LDS (as place_bytes_kernel does): nothing downstream gathers
// windows from the packed reads any more.  The parse's tasks are VIRTUAL tasks here (16 per task, ParseArgs::vt_shift: 640 bins
// instead of 40), so a step takes 16384 records (1024 threads, 32 tiles): a bin's run is ~25 items = 400 bytes, not 6.
// dynamic LDS: u64 cur[nt], u32 tcnt[nt], u32 tpre[nt], u64 srt[PLACE_ITEM_REC], u32 words[PLACE_ITEM_WORDS]
constexpr int PLACE_ITEM_THREADS = 1024;
#ifndef PLACE_ITEM_RPT
#define PLACE_ITEM_RPT 16
#endif
constexpr u32 PLACE_ITEM_REC = PLACE_ITEM_RPT * 1024;  // records of one step (rec_cap * place_group)
constexpr u32 PLACE_ITEM_TILES = PLACE_ITEM_REC / 512; // tiles per step at most
constexpr u32 PLACE_ITEM_WORDS = PLACE_ITEM_TILES * (PARSE_TILE / 16) + 8;      // their packed words + the reach of the last supermer's second word
__global__ __launch_bounds__(PLACE_ITEM_THREADS) void place_items_kernel(ParseArgs a)
{
    // pipelined ingest (place_one): the placement of a slab is launched before the host has seen the scan's verdict.  A tile beyond the record
    // capacity means that the supermers counted (all of them) may exceed the store (rec_cap per tile): nothing is placed, the host falls back.
    if (a.place_one && a.overflow && *(const volatile u32 *)a.overflow) return;
    constexpr int RPT = PLACE_ITEM_REC / PLACE_ITEM_THREADS;
    constexpr int T = PLACE_ITEM_THREADS;
    __shared__ u32 s_w[16];
    __shared__ u32 s_go[PLACE_ITEM_TILES + 4];
    extern __shared__ __attribute__((aligned(16))) u64 s_cur[];
    const int tid = threadIdx.x;
    const u32 nt = a.ntasks;
    u32 *s_tcnt = reinterpret_cast<u32 *>(s_cur + nt);
    u32 *s_tpre = s_tcnt + nt;
    u64 *s_srt = reinterpret_cast<u64 *>(s_tpre + nt + (nt & 1u));
    u32 *s_words = reinterpret_cast<u32 *>(s_srt + PLACE_ITEM_REC);
    for (u32 t = tid; t < nt; t += T) s_cur[t] = a.blk_base[((u64)blockIdx.x * nt + t) * 2];
    const u32 G = a.place_group < PLACE_ITEM_TILES ? a.place_group : PLACE_ITEM_TILES;
    const u32 *p32 = reinterpret_cast<const u32 *>(a.packed);
    const u32 nsl = a.nslabs > 1 ? a.nslabs : 1;
    const int lane = lane_id(), wv = tid >> 6;
    for (u32 sl = a.place_one ? a.slab : 0; sl < (a.place_one ? a.slab + 1 : nsl); ++sl) {
    const u64 tile0 = (u64)sl * a.slab_tiles + (u64)blockIdx.x * a.tiles_per_block;
    const u64 tile_end = (nsl > 1 && ((u64)sl + 1) * a.slab_tiles < a.ntiles) ? ((u64)sl + 1) * a.slab_tiles : a.ntiles;
    for (u32 t0 = 0; t0 < a.tiles_per_block; t0 += G) {
        const u64 tfirst = tile0 + t0;
        if (tfirst >= tile_end) break;
        u32 ng = a.tiles_per_block - t0; if (ng > G) ng = G;
        if (tfirst + ng > tile_end) ng = (u32)(tile_end - tfirst);
        __syncthreads();                                                // previous step done with s_go / s_tcnt / s_srt / s_words
        if (tid < WAVE) {                                               // record offsets of the step's tiles (one wave: 32 tiles)
            u32 n = 0;
            if ((u32)tid < ng) { n = a.tile_nrec[tfirst + tid]; n = n < a.rec_cap ? n : a.rec_cap; }
            const u32 inc = wave_incl_scan(n);
            if ((u32)tid < ng) s_go[tid] = inc - n;
            const u32 tot = __shfl(inc, WAVE - 1);
            if ((u32)tid >= ng && (u32)tid <= PLACE_ITEM_TILES) s_go[tid] = tot;
        }
        for (u32 t = tid; t < nt; t += T) s_tcnt[t] = 0;
        {   // the packed words of the step's tiles (+ reach) as big-endian words
            const u64 w0 = tfirst * (PARSE_TILE / 16);
            const u32 nw = ng * (PARSE_TILE / 16) + 8;
            for (u32 i = tid; i < nw; i += T) {
                const u64 b = (w0 + i) * 4;
                u32 wv2 = 0;
                if (b + 4 <= a.packed_bytes) wv2 = __builtin_bswap32(p32[w0 + i]);
                else if (b < a.packed_bytes) { for (u64 q = b; q < a.packed_bytes; ++q) wv2 |= (u32)a.packed[q] << (24 - 8 * (q & 3)); }
                s_words[i] = wv2;
            }
        }
        __syncthreads();
        const u32 total = s_go[PLACE_ITEM_TILES];
        // a record here: position in tile (11 bits) | k-mers - 1 (4) << 11 | tile of the step (5) << 15 | task (10) << 20
        u32 rec[RPT], rnk[RPT], sub[RPT];
#pragma unroll
        for (int x = 0; x < RPT; ++x) {
            const u32 i = x * T + tid;
            rec[x] = 0xFFFFFFFFu; rnk[x] = 0; sub[x] = 0;
            if (i < total) {
                u32 lo = 0, hi = ng;                                    // the tile of record i: last j with s_go[j] <= i
                while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_go[mid] <= i) lo = mid; else hi = mid; }
                const u64 at = (tfirst + lo) * (u64)a.rec_cap + (i - s_go[lo]);
                const u32 r = a.tile_rec[at];
                sub[x] = a.tile_sub[at];
                const u32 d = (r >> 18) & 1023u;
                rec[x] = (r & 2047u) | (((r >> 11) & 15u) << 11) | (lo << 15) | (d << 20);
                rnk[x] = atomicAdd(&s_tcnt[d], 1u);
            }
        }
        __syncthreads();
        {   // exclusive prefix of the per-task record counts (one lane per task)
            const u32 cnt = (u32)tid < nt ? s_tcnt[tid] : 0u;
            const u32 inc = wave_incl_scan(cnt);
            if (lane == WAVE - 1) s_w[wv] = inc;
            __syncthreads();
            u32 base = 0;
            for (int i = 0; i < 16; ++i) if (i < wv) base += s_w[i];
            if ((u32)tid < nt) s_tpre[tid] = base + inc - cnt;
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < RPT; ++x)
            if (rec[x] != 0xFFFFFFFFu) s_srt[s_tpre[rec[x] >> 20] + rnk[x]] = (u64)rec[x] | ((u64)sub[x] << 32);
        __syncthreads();
        for (u32 i = tid; i < total; i += T) {
            const u64 e = s_srt[i];
            const u32 r = (u32)e;
            const u32 d = r >> 20;
            const u64 slot = s_cur[d] + (i - s_tpre[d]);
            const u32 bit0 = 2u * (((r >> 15) & 31u) * (u32)PARSE_TILE + (r & 2047u));
            const u64 w0 = bits64_be32(s_words, bit0), w1 = bits64_be32(s_words, bit0 + 64u);
            reinterpret_cast<ulonglong2 *>(a.sm_item)[slot] = make_ulonglong2(w0, (w1 & ~0xFFULL) | (u64)(((r >> 11) & 15u) + 1u));
            a.sm_sub[slot] = (u32)(e >> 32);
        }
        __syncthreads();
        for (u32 t = tid; t < nt; t += T) s_cur[t] += s_tcnt[t];
    }
    }                                                                   // slabs
}

// Byte-store placement.  Same job as place_kernel, and in addition every supermer's bases are copied out of the packed
// reads (staged in LDS for the tiles of the step, read ONCE and in order by the whole grid) into the task's byte run.
// A record takes its slot AND its byte range with ONE 64-bit LDS atomic per task ({records << 32 | bytes}), so slot order
// and byte order agree inside a (step, task) run: the byte run of a task is the concatenation of its supermers in slot
// order (what the exchange ships), and sm_boff[slot] points at each one (what the expand on this GPU uses: no prefix sums).
// The bytes leave as unaligned 8-byte stores from one lane per supermer: first 8-byte words front to back, then one word
// ending at the supermer's last byte (overlapping the previous one); consecutive lanes hold consecutive supermers of a
// task, so a wave instruction covers one contiguous stretch.  gfx950 global stores need no alignment (the kernel-mode
// driver runs every queue in unaligned-access mode); a word that straddles a cache line is split by the hardware.
// dynamic LDS: u64 cur[nt], u64 curb[nt], u64 tbase[nt], u64 tc[nt], u32 tpre[nt], u32 pad[nt], u64 srt[PLACE_BYTES_REC], u32 words[...]
// A step takes 8192 records (1024 threads, 16 tiles): with the 320 tasks of eight ranks a (step, task) run is ~25 supermers = 250 bytes (4096 records:
// half that; measured in DESIGN.md 3.2f).
constexpr int PLACE_BYTES_THREADS = 1024;
constexpr u32 PLACE_BYTES_REC = 8192;                  // records of one step (rec_cap * place_group)
constexpr u32 PLACE_BYTES_TILES = 16;                  // tiles per step at most (a record's tile: 4 bits)
constexpr u32 PLACE_BYTES_WORDS = PLACE_BYTES_TILES * (PARSE_TILE / 16) + 24;   // their packed words + the overhang of the last supermer (<= 222 bases) + overread
__global__ __launch_bounds__(PLACE_BYTES_THREADS) void place_bytes_kernel(ParseArgs a)
{
    constexpr int PARSE_THREADS = PLACE_BYTES_THREADS;      // (this kernel's own width)
    constexpr int RPT = PLACE_BYTES_REC / PARSE_THREADS;
    __shared__ u32 s_w[16];
    __shared__ u32 s_go[PLACE_BYTES_TILES + 4];
    extern __shared__ __attribute__((aligned(16))) u64 s_cur[];
    const int tid = threadIdx.x;
    const int K = a.k;
    const u32 nt = a.ntasks;
    u64 *s_curb = s_cur + nt, *s_tbase = s_curb + nt, *s_tc = s_tbase + nt;
    u32 *s_tpre = reinterpret_cast<u32 *>(s_tc + nt);
    u64 *s_srt = reinterpret_cast<u64 *>(s_tpre + 2 * nt);
    u32 *s_words = reinterpret_cast<u32 *>(s_srt + PLACE_BYTES_REC);
    unsigned short *s_sub = reinterpret_cast<unsigned short *>(s_words + PLACE_BYTES_WORDS);      // (only with a.sm_sub16: the host adds the room)
    for (u32 t = tid; t < nt; t += PARSE_THREADS) {
        s_cur[t] = a.blk_base[((u64)blockIdx.x * nt + t) * 2];
        s_curb[t] = a.blk_base[((u64)blockIdx.x * nt + t) * 2 + 1];
        s_tbase[t] = a.task_base3[3 * t + 1];
    }
    const u32 G = a.place_group;
    const u32 *p32 = reinterpret_cast<const u32 *>(a.packed);
    const u32 nsl = a.nslabs > 1 ? a.nslabs : 1;
    for (u32 sl = 0; sl < nsl; ++sl) {
    const u64 tile0 = (u64)sl * a.slab_tiles + (u64)blockIdx.x * a.tiles_per_block;
    const u64 tile_end = (nsl > 1 && ((u64)sl + 1) * a.slab_tiles < a.ntiles) ? ((u64)sl + 1) * a.slab_tiles : a.ntiles;
    for (u32 t0 = 0; t0 < a.tiles_per_block; t0 += G) {
        const u64 tfirst = tile0 + t0;
        if (tfirst >= tile_end) break;
        u32 ng = a.tiles_per_block - t0; if (ng > G) ng = G;
        if (tfirst + ng > tile_end) ng = (u32)(tile_end - tfirst);
        __syncthreads();                                                // previous step done with s_go / s_tc / s_srt / s_words
        if (tid < WAVE) {                                               // record offsets of the step's tiles (one wave)
            u32 n = 0;
            if ((u32)tid < ng) { n = a.tile_nrec[tfirst + tid]; n = n < a.rec_cap ? n : a.rec_cap; }
            const u32 inc = wave_incl_scan(n);
            if ((u32)tid < ng) s_go[tid] = inc - n;
            const u32 tot = __shfl(inc, WAVE - 1);
            if ((u32)tid >= ng && (u32)tid <= PLACE_BYTES_TILES) s_go[tid] = tot;
        }
        for (u32 t = tid; t < nt; t += PARSE_THREADS) s_tc[t] = 0;
        {   // the packed words of the step's tiles (+ overhang) as big-endian words
            const u64 w0 = tfirst * (PARSE_TILE / 16);
            const u32 nw = ng * (PARSE_TILE / 16) + 24;
            for (u32 i = tid; i < nw; i += PARSE_THREADS) {
                const u64 b = (w0 + i) * 4;
                u32 wv = 0;
                if (b + 4 <= a.packed_bytes) wv = __builtin_bswap32(p32[w0 + i]);
                else if (b < a.packed_bytes) { for (u64 q = b; q < a.packed_bytes; ++q) wv |= (u32)a.packed[q] << (24 - 8 * (q & 3)); }
                s_words[i] = wv;
            }
        }
        __syncthreads();
        const u32 total = s_go[PLACE_BYTES_TILES];
        u32 rec[RPT]; u64 got[RPT]; unsigned short sb16[RPT];
#pragma unroll
        for (int x = 0; x < RPT; ++x) {
            const u32 i = x * PARSE_THREADS + tid;
            rec[x] = 0xFFFFFFFFu; got[x] = 0; sb16[x] = 0;
            if (i < total) {
                u32 j = 0, hi = ng;                                     // the tile of record i: last j with s_go[j] <= i
                while (hi - j > 1) { const u32 mid = (j + hi) >> 1; if (s_go[mid] <= i) j = mid; else hi = mid; }
                const u32 r = a.tile_rec[(tfirst + j) * (u64)a.rec_cap + (i - s_go[j])];
                if (a.sm_sub16) sb16[x] = (unsigned short)(a.tile_sub[(tfirst + j) * (u64)a.rec_cap + (i - s_go[j])] >> 16);
                rec[x] = r | (j << 28);
                const u32 nb = (((r >> 11) & 127u) + (u32)K + 3u) >> 2;          // (k-mers - 1) + K bases
                got[x] = atomicAdd((unsigned long long *)&s_tc[(r >> 18) & 1023u], (1ULL << 32) | (unsigned long long)nb);
            }
        }
        __syncthreads();
        {   // exclusive prefix of the per-task record counts (one lane per task)
            const u32 cnt = (u32)tid < nt ? (u32)(s_tc[tid] >> 32) : 0u;
            const u32 inc = wave_incl_scan(cnt);
            if (lane_id() == WAVE - 1) s_w[tid >> 6] = inc;
            __syncthreads();
            u32 base = 0;
            for (int i = 0; i < 16; ++i) if (i < (tid >> 6)) base += s_w[i];
            if ((u32)tid < nt) s_tpre[tid] = base + inc - cnt;
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < RPT; ++x)
            if (rec[x] != 0xFFFFFFFFu) {
                const u32 at = s_tpre[(rec[x] >> 18) & 1023u] + (u32)(got[x] >> 32);
                s_srt[at] = (u64)rec[x] | (got[x] << 32);   // {record, byte offset in the task's run}
                if (a.sm_sub16) s_sub[at] = sb16[x];
            }
        __syncthreads();
        for (u32 i = tid; i < total; i += PARSE_THREADS) {
            const u64 e = s_srt[i];
            const u32 r = (u32)e, bo = (u32)(e >> 32);
            const u32 d = (r >> 18) & 1023u;
            if (a.task_skip && a.task_skip[d]) continue;
            const u64 slot = s_cur[d] + (i - s_tpre[d]);
            const u64 babs = s_curb[d] + bo;
            const u32 len = ((r >> 11) & 127u) + (u32)K;
            const u32 nb = (len + 3) >> 2;
            a.sm_len[slot] = (u8)len;
            if (a.sm_boff) a.sm_boff[slot] = (u32)(babs - s_tbase[d]);
            if (a.sm_sub16) a.sm_sub16[slot] = s_sub[i];
            if (a.sm_gpos) a.sm_gpos[slot] = (tfirst + (r >> 28)) * PARSE_TILE + (u64)(r & 2047);
            const u32 bit0 = 2u * ((r >> 28) * (u32)PARSE_TILE + (r & 2047u));
            const u64 tailmask = (len & 3u) ? ~(u64)(0xFFu >> (2 * (len & 3u))) : ~0ULL;   // zero bits behind the last base (in its byte = the lowest byte of a big-endian word)
            u8 *dst = a.sm_bytes + babs;
            if (nb >= 8) {
                const u32 nfull = nb >> 3, rem = nb & 7u;
                for (u32 q = 0; q < nfull; ++q) {
                    u64 v = bits64_be32(s_words, bit0 + 64u * q);
                    if (rem == 0 && q + 1 == nfull) v &= tailmask;
                    *reinterpret_cast<u64 *>(dst + 8u * q) = __builtin_bswap64(v);
                }
                if (rem) { const u64 v = bits64_be32(s_words, bit0 + 8u * (nb - 8u)) & tailmask; *reinterpret_cast<u64 *>(dst + (nb - 8u)) = __builtin_bswap64(v); }
            } else {
                const u64 v = bits64_be32(s_words, bit0) & (tailmask << (8u * (8u - nb)) | ~(~0ULL >> (8u * (nb - 1u))));
                for (u32 q = 0; q < nb; ++q) dst[q] = (u8)(v >> (56u - 8u * q));
            }
        }
        __syncthreads();
        for (u32 t = tid; t < nt; t += PARSE_THREADS) { const u64 c = s_tc[t]; s_cur[t] += c >> 32; s_curb[t] += c & 0xFFFFFFFFu; }
    }
    }                                                                   // slabs
}

// EXTENSION: (PosInRead, ReadId) of every supermer from its base position (one index search per supermer;
// the reference carries them in length_t, include/kmer.hpp:350-360)
__device__ __forceinline__ u64 s_gpos_load(const u64 *p, u64 i) { return p[i]; }
__global__ void resolve_pos_rid_kernel(const u64 *sm_gpos, u64 n, const u64 *roff, u64 nreads, int64_t rid_base, u32 *sm_pos, int32_t *sm_rid,
                                       const u32 *tile_r0, u64 ntiles)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
        const u64 g = s_gpos_load(sm_gpos, s);
        // with the per-tile hint the read is among the few that start inside the supermer's 2048-position tile
        u64 lo = 0, hi = nreads - 1;
        if (tile_r0) {
            const u64 tile = g / PARSE_TILE;
            lo = tile_r0[tile];
            hi = (tile + 1 < ntiles) ? tile_r0[tile + 1] : nreads - 1;      // the next tile's first read starts at or before that tile's first byte
            if (hi < lo) hi = lo;
        }
        const u64 r = find_read(roff, lo, hi, g >> 2);
        sm_pos[s] = (u32)(g - roff[r] * 4);
        sm_rid[s] = (int32_t)(rid_base + (int64_t)r);
    }
}

// the reads that lie completely inside the first `bytes` bytes of the packed buffer: out[0] = their number r, out[1] = roff[r] (plan estimate, hsk_api.hip)
__global__ void prefix_reads_kernel(const u64 *roff, u64 nreads, u64 bytes, u64 *out)
{
    if (threadIdx.x || blockIdx.x) return;
    const u64 r = nreads ? find_read(roff, 0, nreads - 1, bytes) : 0;      // the read that holds byte `bytes` (or the last one that starts at or before it)
    out[0] = r; out[1] = nreads ? roff[r] : 0;
}

// hsk_count(): the caller's read index must be ascending, non-overlapping and inside the packed buffer (error bit 32)
__global__ void index_check_kernel(const u64 *roff, const u32 *rlen, u64 nreads, u64 packed_bytes, u32 *err)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    bool bad = false;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
        const u64 o = roff[r], end = o + ((u64)rlen[r] + 3) / 4;
        if (end > packed_bytes || end < o) bad = true;
        if (r == 0 ? o != 0 : o < roff[r - 1] + ((u64)rlen[r - 1] + 3) / 4) bad = true;
    }
    if (bad) atomicOr(err, 32u);
}

// ---- read offsets derived from the read lengths (hsk_count with a pinned DnaBuffer) ----------------------------------------
// A DnaBuffer stores its reads back to back, every read on a byte boundary (reference src/dnabuffer.cpp:24-31), so
// read_byte_off[r] = sum of (len + 3) / 4 over the reads before r.  Only the lengths (4 B per read) are copied ahead of the
// scan; the caller's offsets (8 B per read) do not travel at all: host threads compare them with the same prefix sums while
// the GPU scans (hsk_api.hip: offsets_back_to_back).  A buffer with gaps between its reads is counted again with the caller's offsets.
// read lengths that are all the same (fixed-length short reads): generated on the device instead of copied (hsk_count(), derive_input)
__global__ void rlen_fill_kernel(u32 *rlen, u64 nreads, u32 len)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) rlen[r] = len;
}

constexpr int ROFF_TILE = PARSE_THREADS * 8;
__global__ __launch_bounds__(PARSE_THREADS) void roff_tilesum_kernel(const u32 *rlen, u64 nreads, u64 *tile_sum)
{
    __shared__ u64 s_scr[8];
    const u64 base = (u64)blockIdx.x * ROFF_TILE + (u64)threadIdx.x * 8;
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (base + i < nreads) sum += ((u64)rlen[base + i] + 3) >> 2;
    u64 tot;
    (void)block_excl_scan_256<u64>(sum, s_scr, &tot);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
// exclusive scan of the tile sums in place; one workgroup, 8 tiles per lane and step
__global__ __launch_bounds__(PARSE_THREADS) void roff_tilescan_kernel(u64 *tile_sum, u64 ntiles)
{
    __shared__ u64 s_scr[8];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < ntiles; base += (u64)PARSE_THREADS * 8) {
        const u64 t0 = base + (u64)threadIdx.x * 8;
        u64 v[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = t0 + i < ntiles ? tile_sum[t0 + i] : 0; sum += v[i]; }
        u64 tot;
        u64 ex = block_excl_scan_256<u64>(sum, s_scr, &tot) + s_carry;
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (t0 + i < ntiles) tile_sum[t0 + i] = ex; ex += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
}
__global__ __launch_bounds__(PARSE_THREADS) void roff_write_kernel(const u32 *rlen, u64 nreads, const u64 *tile_off, u64 *roff)
{
    __shared__ u64 s_scr[8];
    const u64 base = (u64)blockIdx.x * ROFF_TILE + (u64)threadIdx.x * 8;
    u64 nb[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { nb[i] = base + i < nreads ? ((u64)rlen[base + i] + 3) >> 2 : 0; sum += nb[i]; }
    u64 ex = block_excl_scan_256<u64>(sum, s_scr, nullptr) + tile_off[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (base + i < nreads) roff[base + i] = ex; ex += nb[i]; }
}
// column sums of the COUNT matrix for task t; eight rows are requested before the first is added (one thread walks
// ~1000 rows: a load per step would be a memory latency per row)
__device__ __forceinline__ void col_sums(const u64 *blk_cnt, u32 nblocks, u32 ntasks, u32 t, u64 &s, u64 &b, u64 &k)
{
    constexpr int U = 8;
    u32 blk = 0;
    for (; blk + U <= nblocks; blk += U) {
        u64 v[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) { const u64 *c = blk_cnt + ((u64)(blk + u) * ntasks + t) * 3; v[u][0] = c[0]; v[u][1] = c[1]; v[u][2] = c[2]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { s += v[u][0]; b += v[u][1]; k += v[u][2]; }
    }
    for (; blk < nblocks; ++blk) { const u64 *c = blk_cnt + ((u64)blk * ntasks + t) * 3; s += c[0]; b += c[1]; k += c[2]; }
}

// task_tot[t][3] = column sums of the COUNT matrix (supermers, bytes, k-mers of task t on this rank)
__global__ void task_totals_kernel(const u64 *blk_cnt, u32 nblocks, u32 ntasks, u64 *task_tot)
{
    const u32 t = threadIdx.x;
    if (t >= ntasks) return;
    u64 s = 0, b = 0, k = 0;
    col_sums(blk_cnt, nblocks, ntasks, t, s, b, k);
    task_tot[3 * t] = s; task_tot[3 * t + 1] = b; task_tot[3 * t + 2] = k;
}

// Exclusive scan of the COUNT matrix: per task totals, task bases (tasks laid out in `order`),
// and per (block, task) cursors for EMIT.  One thread per task; the matrix is tiny.
//   task_tot[t][3], task_base[t][3] (supermer slot, byte, kmer), blk_base[b][t][2]
// run (optional, [3]): the slot / byte / k-mer totals of everything laid out before this call (slabs placed one by one): the bases
// start there and the totals of this call are added to it.
__global__ void parse_scan_kernel(const u64 *blk_cnt, u32 nblocks, u32 ntasks, const u32 *order, const u8 *skip,
                                  u64 *task_tot, u64 *task_base, u64 *blk_base, u64 *run = nullptr)
{
    __shared__ u64 s_tot[HSK_MAX_TASKS * 3];
    const u32 t = threadIdx.x;
    const bool live = t < ntasks && !(skip && skip[t]);      // skipped tasks take no room
    if (t < ntasks) {
        u64 s = 0, b = 0, k = 0;
        if (live) col_sums(blk_cnt, nblocks, ntasks, t, s, b, k);
        s_tot[3 * t] = s; s_tot[3 * t + 1] = b; s_tot[3 * t + 2] = k;
        task_tot[3 * t] = s; task_tot[3 * t + 1] = b; task_tot[3 * t + 2] = k;
    }
    __syncthreads();
    {
        // bases of the tasks in storage order: an exclusive scan over order[] (one lane per position; the virtual tasks of the combining
        // extraction make this 640 long, and a serial walk of it was a fifth of the kernel)
        __shared__ u64 s_part[3][16];
        const u32 task = t < ntasks ? order[t] : 0u;
        u64 v0 = t < ntasks ? s_tot[3 * task] : 0, v1 = t < ntasks ? s_tot[3 * task + 1] : 0, v2 = t < ntasks ? s_tot[3 * task + 2] : 0;
        const int lane = lane_id(), w = (int)(t >> 6);
        const u64 i0 = wave_incl_scan(v0), i1 = wave_incl_scan(v1), i2 = wave_incl_scan(v2);
        if (lane == WAVE - 1) { s_part[0][w] = i0; s_part[1][w] = i1; s_part[2][w] = i2; }
        __syncthreads();
        u64 b0 = run ? run[0] : 0, b1 = run ? run[1] : 0, b2 = run ? run[2] : 0, t0 = 0, t1 = 0, t2 = 0;
        for (int i = 0; i < 16; ++i) {
            const u64 p0 = s_part[0][i], p1 = s_part[1][i], p2 = s_part[2][i];
            if (i < w) { b0 += p0; b1 += p1; b2 += p2; }
            t0 += p0; t1 += p1; t2 += p2;
        }
        if (t < ntasks) { task_base[3 * task] = b0 + i0 - v0; task_base[3 * task + 1] = b1 + i1 - v1; task_base[3 * task + 2] = b2 + i2 - v2; }
        __syncthreads();                                       // (everybody has read run[] before it moves)
        if (t == 0 && run) { run[0] += t0; run[1] += t1; run[2] += t2; }
    }
    __syncthreads();
    if (t < ntasks) {
        u64 s = task_base[3 * t], b = task_base[3 * t + 1];
        constexpr int U = 16;
        u32 blk = 0;
        for (; blk + U <= nblocks; blk += U) {
            u64 v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) { const u64 *c = blk_cnt + ((u64)(blk + u) * ntasks + t) * 3; v[u][0] = c[0]; v[u][1] = c[1]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                u64 *o = blk_base + ((u64)(blk + u) * ntasks + t) * 2;
                o[0] = s; o[1] = b;
                if (live) { s += v[u][0]; b += v[u][1]; }
            }
        }
        for (; blk < nblocks; ++blk) {
            const u64 *c = blk_cnt + ((u64)blk * ntasks + t) * 3;
            u64 *o = blk_base + ((u64)blk * ntasks + t) * 2;
            o[0] = s; o[1] = b;
            if (live) { s += c[0]; b += c[1]; }
        }
    }
}

// The same scan for many tasks (the combining extraction's virtual tasks: 640 columns, and one call per ingest slab): parse_scan_kernel
// is ONE workgroup whose lanes walk ~1000 rows twice, 0.4 ms.  Here the rows are cut into PS_SEGS segments: (1) every (segment, task)
// lane sums its rows, (2) one workgroup turns the sums into task totals and task bases (block scan over order[]), (3) every (segment,
// task) lane walks its rows again and writes the cursors.  Same outputs, ~30 us.
constexpr u32 PS_SEGS = 16;
constexpr u32 PS_THREADS = 256;
// part[seg][task][3]
__global__ __launch_bounds__(PS_THREADS) void parse_scan_part_kernel(const u64 *blk_cnt, u32 nblocks, u32 ntasks, const u8 *skip, u64 *part)
{
    const u32 t = blockIdx.x * PS_THREADS + threadIdx.x, seg = blockIdx.y;
    if (t >= ntasks) return;
    const u32 per = (nblocks + PS_SEGS - 1) / PS_SEGS;
    const u32 b0 = seg * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    u64 s = 0, b = 0, k = 0;
    if (!(skip && skip[t]) && b0 < b1) col_sums(blk_cnt + (u64)b0 * ntasks * 3, b1 - b0, ntasks, t, s, b, k);
    u64 *o = part + ((u64)seg * ntasks + t) * 3;
    o[0] = s; o[1] = b; o[2] = k;
}
__global__ __launch_bounds__(HSK_MAX_TASKS) void parse_scan_base_kernel(const u64 *part, u32 ntasks, const u32 *order, u64 *task_tot, u64 *task_base, u64 *run)
{
    __shared__ u64 s_tot[HSK_MAX_TASKS * 3];
    __shared__ u64 s_part[3][16];
    const u32 t = threadIdx.x;
    if (t < ntasks) {
        u64 s = 0, b = 0, k = 0;
        for (u32 g = 0; g < PS_SEGS; ++g) { const u64 *p = part + ((u64)g * ntasks + t) * 3; s += p[0]; b += p[1]; k += p[2]; }
        s_tot[3 * t] = s; s_tot[3 * t + 1] = b; s_tot[3 * t + 2] = k;
        task_tot[3 * t] = s; task_tot[3 * t + 1] = b; task_tot[3 * t + 2] = k;
    }
    __syncthreads();
    const u32 task = t < ntasks ? order[t] : 0u;
    const u64 v0 = t < ntasks ? s_tot[3 * task] : 0, v1 = t < ntasks ? s_tot[3 * task + 1] : 0, v2 = t < ntasks ? s_tot[3 * task + 2] : 0;
    const int lane = lane_id(), w = (int)(t >> 6);
    const u64 i0 = wave_incl_scan(v0), i1 = wave_incl_scan(v1), i2 = wave_incl_scan(v2);
    if (lane == WAVE - 1) { s_part[0][w] = i0; s_part[1][w] = i1; s_part[2][w] = i2; }
    __syncthreads();
    u64 b0 = run ? run[0] : 0, b1 = run ? run[1] : 0, b2 = run ? run[2] : 0, t0 = 0, t1 = 0, t2 = 0;
    for (int i = 0; i < 16; ++i) {
        const u64 p0 = s_part[0][i], p1 = s_part[1][i], p2 = s_part[2][i];
        if (i < w) { b0 += p0; b1 += p1; b2 += p2; }
        t0 += p0; t1 += p1; t2 += p2;
    }
    if (t < ntasks) { task_base[3 * task] = b0 + i0 - v0; task_base[3 * task + 1] = b1 + i1 - v1; task_base[3 * task + 2] = b2 + i2 - v2; }
    __syncthreads();
    if (t == 0 && run) { run[0] += t0; run[1] += t1; run[2] += t2; }
}
__global__ __launch_bounds__(PS_THREADS) void parse_scan_fill_kernel(const u64 *blk_cnt, u32 nblocks, u32 ntasks, const u8 *skip, const u64 *part, const u64 *task_base, u64 *blk_base)
{
    const u32 t = blockIdx.x * PS_THREADS + threadIdx.x, seg = blockIdx.y;
    if (t >= ntasks) return;
    const bool live = !(skip && skip[t]);
    const u32 per = (nblocks + PS_SEGS - 1) / PS_SEGS;
    const u32 b0 = seg * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    u64 s = task_base[3 * t], b = task_base[3 * t + 1];
    for (u32 g = 0; g < seg; ++g) { const u64 *p = part + ((u64)g * ntasks + t) * 3; s += p[0]; b += p[1]; }
    constexpr int U = 8;
    u32 blk = b0;
    for (; blk + U <= b1; blk += U) {
        u64 v[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) { const u64 *c = blk_cnt + ((u64)(blk + u) * ntasks + t) * 3; v[u][0] = c[0]; v[u][1] = c[1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u64 *o = blk_base + ((u64)(blk + u) * ntasks + t) * 2;
            o[0] = s; o[1] = b;
            if (live) { s += v[u][0]; b += v[u][1]; }
        }
    }
    for (; blk < b1; ++blk) {
        const u64 *c = blk_cnt + ((u64)blk * ntasks + t) * 3;
        u64 *o = blk_base + ((u64)blk * ntasks + t) * 2;
        o[0] = s; o[1] = b;
        if (live) { s += c[0]; b += c[1]; }
    }
}

} // namespace hsk
