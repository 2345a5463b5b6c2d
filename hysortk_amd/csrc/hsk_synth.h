// hsk_synth.h -- synthetic reads generated directly in HBM (bench / tests only).
// S-reads(G, c) of BASELINE.md: a random genome of G bases and error-free fixed-length reads
// sampled uniformly from it, strand 50/50.  The numpy twin of this generator is
// hysortk_amd/synth.py (same splitmix64 streams), so the oracle can be fed identical reads.
#pragma once
#include "hsk_device.h"

namespace hsk {

// genome word j holds bases 32j .. 32j+31, base i at bits 2*(i%32) (low first)
__global__ void synth_genome_kernel(u64 *gw, u64 nwords, u64 seed)
{
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nwords) gw[j] = splitmix64(seed * 0x9e3779b97f4a7c15ULL + j);
}

__device__ __forceinline__ u32 genome_base(const u64 *gw, u64 i) { return (u32)(gw[i >> 5] >> (2 * (i & 31))) & 3u; }

// one thread per output byte; read r occupies bytes [r*nb, (r+1)*nb)
// err_thresh: substitution errors, one per base with probability err_thresh / 2^32 (a base is replaced by one of the other
// three: sequencing errors make most erroneous k-mers singletons, which is what stresses the aggregation's hash tables)
__global__ void synth_reads_kernel(const u64 *gw, u64 genome_len, u32 read_len, u64 nreads, u64 seed2, u8 *packed, u32 err_thresh = 0)
{
    const u32 nb = (read_len + 3) >> 2;
    // (a grid of one lane per byte ends at 2^32 lanes = 17 Gbp: the lanes stride)
    for (u64 byte = (u64)blockIdx.x * blockDim.x + threadIdx.x; byte < nreads * nb; byte += (u64)gridDim.x * blockDim.x) {
    const u64 r = byte / nb;
    const u32 jb = (u32)(byte - r * nb);
    const u64 h = splitmix64(seed2 + r);
    const u64 start = (h >> 1) % (genome_len - read_len + 1);
    const bool rc = h & 1;
    u32 out = 0;
    for (u32 b = 0; b < 4; ++b) {
        const u32 j = jb * 4 + b;
        if (j >= read_len) break;
        u32 base = rc ? (3u - genome_base(gw, start + read_len - 1 - j)) : genome_base(gw, start + j);
        if (err_thresh) {
            const u64 e = splitmix64((h ^ 0x5bd1e995ULL) + (u64)j * 0x9e3779b97f4a7c15ULL);
            if ((u32)e < err_thresh) base = (base + 1u + (u32)((e >> 32) % 3u)) & 3u;
        }
        out |= base << (6 - 2 * b);
    }
    packed[byte] = (u8)out;
    }
}

__global__ void synth_index_kernel(u64 *roff, u32 *rlen, u64 nreads, u32 read_len)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nb = (read_len + 3) >> 2;
    if (r <= nreads) roff[r] = r * nb;
    if (r < nreads) rlen[r] = read_len;
}

// FASTA text -> DnaBuffer bytes on the device (what DnaSeq::compress does on the host, reference src/dnaseq.cpp:9-31,
// fed by FastaIndex's line arithmetic, src/fastaindex.cpp:285: base i of a record sits at
// pos + (i / linebases) * linewidth + i % linebases).  One lane per output byte; the record of a byte is found by
// binary search over the output offsets.  Codes A/a/N/n = 0, C/c = 1, G/g = 2, T/t = 3; any other character gets the
// reference's code 4, OR-ed in at the base's shift and truncated to the byte exactly as `uint8_t |= 4 << shift` does.
__global__ void pack_fasta_kernel(const u8 *text, u64 text_bytes, const u64 *rec_pos, const u32 *rec_len, const u32 *line_bases, const u32 *line_width,
                                  const u64 *roff, u64 nrec, u64 out_bytes, u8 *out)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < out_bytes; b += stride) {
        u64 lo = 0, hi = nrec - 1;                                 // last record whose first output byte is <= b
        while (lo < hi) { const u64 mid = lo + (hi - lo + 1) / 2; if (roff[mid] <= b) lo = mid; else hi = mid - 1; }
        const u64 r = lo;
        const u32 len = rec_len[r], lb = line_bases[r], lw = line_width[r];
        const u64 pos = rec_pos[r];
        const u64 i0 = (b - roff[r]) * 4;
        u32 byte = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 i = i0 + j;
            if (i >= len) break;
            const u64 src = pos + (lb ? (i / lb) * (u64)lw + (i % lb) : i);
            const u8 ch = src < text_bytes ? text[src] : (u8)'A';
            u32 code;
            switch (ch) {
            case 'A': case 'a': case 'N': case 'n': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': code = 3; break;
            default: code = 4; break;
            }
            byte |= code << (6 - 2 * j);
        }
        out[b] = (u8)byte;
    }
}

// roff[r] = sum of (len + 3) / 4 over the records before r (nrec + 1 values); single workgroup, sequential chunks
__global__ void pack_fasta_offsets_kernel(const u32 *rec_len, u64 nrec, u64 *roff)
{
    __shared__ u64 s_scr[8];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < nrec; base += 256) {
        const u64 r = base + threadIdx.x;
        const u64 nb = r < nrec ? ((u64)rec_len[r] + 3) / 4 : 0;
        u64 tot;
        const u64 e = block_excl_scan_256<u64>(nb, s_scr, &tot);
        if (r < nrec) roff[r] = s_carry + e;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) roff[nrec] = s_carry;
}

// ---- result egress: entries -> the lines write_output_file prints (reference src/hysortk.cpp:138-164: "KMERSTRING\tcount\n",
//      the k-mer spelled by Kmer::GetString, include/kmer.hpp:76).  Two launches around a scan of the tile sizes:
//      line lengths (K + 2 + decimal digits of the count), then every lane writes its own line. --------------------------------
constexpr int FMT_THREADS = 256;

__device__ __forceinline__ u32 dec_digits(u64 v) { u32 d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

// tile_bytes[tile] = text bytes of the tile's entries (COUNT) / exclusive offsets (EMIT, after the scan)
template <bool EMIT>
__global__ __launch_bounds__(FMT_THREADS) void format_entries_kernel(const u64 *entries, u64 n, int nw, int k, u64 *tile_bytes, char *text)
{
    __shared__ u64 s_scr[8];
    const u64 e = (u64)blockIdx.x * FMT_THREADS + threadIdx.x;
    u64 cnt = 0; u32 len = 0;
    if (e < n) { cnt = entries[e * (nw + 1) + nw]; len = (u32)k + 2 + dec_digits(cnt); }
    u64 tot;
    const u64 off = block_excl_scan_256<u64>((u64)len, s_scr, &tot);
    if (!EMIT) { if (threadIdx.x == 0) tile_bytes[blockIdx.x] = tot; return; }
    if (e >= n) return;
    char *o = text + tile_bytes[blockIdx.x] + off;
    for (int j = 0; j < k; ++j) {
        const u64 w = entries[e * (nw + 1) + (j >> 5)];
        o[j] = "ACGT"[(w >> (2 * (31 - (j & 31)))) & 3];
    }
    o[k] = '\t';
    const u32 nd = len - (u32)k - 2;
    u64 v = cnt;
    for (u32 d = 0; d < nd; ++d) { o[k + nd - d] = (char)('0' + (int)(v % 10)); v /= 10; }
    o[k + 1 + nd] = '\n';
}

} // namespace hsk
