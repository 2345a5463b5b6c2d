"""Synthetic reads S-reads(G, c): numpy twin of csrc/hsk_synth.h (same splitmix64
streams), used by tests to feed the oracle the same reads the GPU generates in HBM."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def genome_codes(genome_len, seed):
    """uint8 codes (0..3) of the synthetic genome."""
    nwords = (genome_len + 31) // 32
    with np.errstate(over="ignore"):
        words = splitmix64(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.arange(nwords, dtype=np.uint64))
    shifts = (2 * np.arange(32, dtype=np.uint64))[None, :]
    codes = ((words[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)
    return codes[:genome_len]


def reads(genome_len, read_len, nreads, seed, first_read=0):
    """Returns (list of ASCII reads) identical to what hsk_synth_reads writes (after 2-bit packing)."""
    g = genome_codes(genome_len, seed)
    seed2 = splitmix64(np.uint64(seed) ^ np.uint64(0xABCDEF12345))
    with np.errstate(over="ignore"):
        h = splitmix64(seed2 + np.uint64(first_read) + np.arange(nreads, dtype=np.uint64))
    start = (h >> np.uint64(1)) % np.uint64(genome_len - read_len + 1)
    rc = (h & np.uint64(1)).astype(bool)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for r in range(nreads):
        s = int(start[r])
        seg = g[s:s + read_len]
        if rc[r]:
            seg = (3 - seg)[::-1]
        out.append(lut[seg].tobytes().decode())
    return out


def packed_reads(genome_len, read_len, nreads, seed, first_read=0):
    """Vectorised: (packed uint8[nreads*nb], read_off uint64[nreads], read_len uint32[nreads])."""
    g = genome_codes(genome_len, seed)
    seed2 = splitmix64(np.uint64(seed) ^ np.uint64(0xABCDEF12345))
    with np.errstate(over="ignore"):
        h = splitmix64(seed2 + np.uint64(first_read) + np.arange(nreads, dtype=np.uint64))
    start = ((h >> np.uint64(1)) % np.uint64(genome_len - read_len + 1)).astype(np.int64)
    rc = (h & np.uint64(1)).astype(bool)
    nb = (read_len + 3) // 4
    j = np.arange(nb * 4, dtype=np.int64)[None, :]
    idx_f = start[:, None] + j
    idx_r = start[:, None] + (read_len - 1 - j)
    valid = (j < read_len)
    idx = np.where(rc[:, None], idx_r, idx_f)
    idx = np.clip(idx, 0, genome_len - 1)
    codes = g[idx]
    codes = np.where(rc[:, None], 3 - codes, codes)
    codes = np.where(valid, codes, 0).astype(np.uint8).reshape(nreads, nb, 4)
    packed = (codes[:, :, 0] << 6) | (codes[:, :, 1] << 4) | (codes[:, :, 2] << 2) | codes[:, :, 3]
    off = (np.arange(nreads, dtype=np.uint64) * np.uint64(nb))
    lens = np.full(nreads, read_len, dtype=np.uint32)
    return np.ascontiguousarray(packed.reshape(-1).astype(np.uint8)), off, lens
