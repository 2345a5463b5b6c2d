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
__global__ void synth_reads_kernel(const u64 *gw, u64 genome_len, u32 read_len, u64 nreads, u64 seed2, u8 *packed)
{
    const u32 nb = (read_len + 3) >> 2;
    const u64 byte = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (byte >= nreads * nb) return;
    const u64 r = byte / nb;
    const u32 jb = (u32)(byte - r * nb);
    const u64 h = splitmix64(seed2 + r);
    const u64 start = (h >> 1) % (genome_len - read_len + 1);
    const bool rc = h & 1;
    u32 out = 0;
    for (u32 b = 0; b < 4; ++b) {
        const u32 j = jb * 4 + b;
        if (j >= read_len) break;
        const u32 base = rc ? (3u - genome_base(gw, start + read_len - 1 - j)) : genome_base(gw, start + j);
        out |= base << (6 - 2 * b);
    }
    packed[byte] = (u8)out;
}

__global__ void synth_index_kernel(u64 *roff, u32 *rlen, u64 nreads, u32 read_len)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nb = (read_len + 3) >> 2;
    if (r <= nreads) roff[r] = r * nb;
    if (r < nreads) rlen[r] = read_len;
}

} // namespace hsk
