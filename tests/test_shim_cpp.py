"""The C++ drop-in headers (include/hysortk/): host-side types against the reference's golden
vectors (no GPU), and -- on the GPU box -- the example driver end to end."""
import os
import subprocess

import pytest

from tests import util

CHECK_SRC = r'''
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include "hysortk/hysortk.hpp"
using namespace hysortk;
int main(int argc, char** argv) {
    std::ifstream in(argv[1]);
    std::string s;
    while (std::getline(in, s)) {
        DnaBuffer buf(DnaSeq::bytesneeded(s.size()) + 1);
        buf.push_back(s.data(), s.size());
        DnaBuffer copy(buf);                       // deep copy must re-seat the views
        const DnaSeq& q = copy[0];
        std::printf("P ");
        for (size_t i = 0; i < q.numbytes(); ++i) std::printf("%02x", q.data()[i]);
        std::printf("\nA %s\n", q.ascii().c_str());
        auto reps = TKmer::GetRepKmers(q);
        for (auto& k : reps) {
            const uint64_t* w = (const uint64_t*)k.GetBytes();
            std::printf("K");
            for (int j = 0; j < TKmer::NBYTES / 8; ++j) std::printf(" %016llx", (unsigned long long)w[j]);
            std::printf(" %s %016llx\n", k.GetString().c_str(), (unsigned long long)k.GetHash());
        }
    }
    std::printf("S %zu\n", sizeof(KmerListEntryS));
    return 0;
}
'''


@pytest.mark.parametrize("variant", ["k31", "k51"])
def test_shim_types_against_golden(variant, tmp_path):
    from oracle import hsk_oracle as O
    cfg = util.VARIANTS[variant]
    g = util.load_json("stages_%s.json" % variant)
    reads = [rd for rd in g["reads"] if len(rd["seq"]) > 0]
    (tmp_path / "seqs.txt").write_text("\n".join(rd["seq"] for rd in reads) + "\n")
    (tmp_path / "check.cpp").write_text(CHECK_SRC)
    exe = str(tmp_path / "check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-DKMER_SIZE=%d" % cfg["k"],
                           "-DMINIMIZER_SIZE=%d" % cfg["m"], "-DLOWER_KMER_FREQ=1", "-DUPPER_KMER_FREQ=65535", "-DEXTENSION=0",
                           "-o", exe, str(tmp_path / "check.cpp"), "-L", os.path.join(util.ROOT, "hysortk_amd"), "-lhsk",
                           "-Wl,-rpath," + os.path.join(util.ROOT, "hysortk_amd")])
    out = subprocess.check_output([exe, str(tmp_path / "seqs.txt")]).decode().splitlines()
    it = iter(out)
    nw = (cfg["k"] + 31) // 32
    for rd in reads:
        assert next(it) == "P " + rd["packed"]
        seq = rd["seq"].upper().replace("N", "A")
        assert next(it) == "A " + seq
        for ws in rd["repkmers"]:
            parts = next(it).split()
            assert parts[0] == "K" and parts[1:1 + nw] == ws
            words = util.hex_words(ws)
            assert parts[1 + nw] == util.words_to_str(words, cfg["k"])
            assert int(parts[2 + nw], 16) == O.murmur64(words)          # Kmer::GetHash = murmur over all key bytes
    assert next(it) == "S %d" % (8 * (nw + 1))                          # KmerListEntryS layout = hsk_result.entries


@pytest.mark.gpu
def test_example_driver_end_to_end(tmp_path):
    exe = str(tmp_path / "hysortk")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-DKMER_SIZE=31", "-DMINIMIZER_SIZE=17",
                           "-DLOWER_KMER_FREQ=1", "-DUPPER_KMER_FREQ=65535", "-DEXTENSION=0", "-o", exe,
                           os.path.join(util.ROOT, "examples", "hysortk_main.cpp"), "-L", os.path.join(util.ROOT, "hysortk_amd"), "-lhsk",
                           "-Wl,-rpath," + os.path.join(util.ROOT, "hysortk_amd")])
    outdir = tmp_path / "out"
    outdir.mkdir()
    so = subprocess.check_output([exe, util.GOLDEN + "/reads_small.fa", str(outdir)]).decode()
    assert open(util.GOLDEN + "/hist_k31.txt").read() in so                                # histogram text, byte for byte
    lines = sorted(open(outdir / "0.out").read().splitlines())
    assert lines == sorted("%s\t%d" % (g[0], g[1]) for g in util.load_count("count_k31.txt"))


@pytest.mark.gpu
def test_example_driver_extension(tmp_path):
    exe = str(tmp_path / "hysortk_ext")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-DKMER_SIZE=31", "-DMINIMIZER_SIZE=17",
                           "-DLOWER_KMER_FREQ=1", "-DUPPER_KMER_FREQ=65535", "-DEXTENSION=1", "-o", exe,
                           os.path.join(util.ROOT, "examples", "hysortk_main.cpp"), "-L", os.path.join(util.ROOT, "hysortk_amd"), "-lhsk",
                           "-Wl,-rpath," + os.path.join(util.ROOT, "hysortk_amd")])
    outdir = tmp_path / "out"
    outdir.mkdir()
    subprocess.check_call([exe, util.GOLDEN + "/reads_small.fa", str(outdir)], stdout=subprocess.DEVNULL)
    lines = sorted(open(outdir / "0.out").read().splitlines())
    assert lines == sorted("%s\t%d" % (g[0], g[1]) for g in util.load_count("count_k31ext.txt"))


@pytest.mark.gpu
def test_example_driver_k51(tmp_path):
    """Two-word keys through the C++ shim: -DKMER_SIZE=51, raw list against the reference's K=51 (RADULS) run."""
    exe = str(tmp_path / "hysortk_k51")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-DKMER_SIZE=51", "-DMINIMIZER_SIZE=17",
                           "-DLOWER_KMER_FREQ=1", "-DUPPER_KMER_FREQ=65535", "-DEXTENSION=0", "-o", exe,
                           os.path.join(util.ROOT, "examples", "hysortk_main.cpp"), "-L", os.path.join(util.ROOT, "hysortk_amd"), "-lhsk",
                           "-Wl,-rpath," + os.path.join(util.ROOT, "hysortk_amd")])
    outdir = tmp_path / "out"
    outdir.mkdir()
    so = subprocess.check_output([exe, util.GOLDEN + "/reads_small.fa", str(outdir)]).decode()
    assert open(util.GOLDEN + "/hist_k51.txt").read() in so
    lines = sorted(open(outdir / "0.out").read().splitlines())
    assert lines == sorted("%s\t%d" % (g[0], g[1]) for g in util.load_count("count_k51.txt"))


SURFACE_SRC = r'''
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include "hysortk/hysortk.hpp"
using namespace hysortk;
int main() {
    const std::string a = "ACGTTGCAACGTACGTTTGACCATGACCAGTAGGATTACAGATTACA", b = "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTACGT";
    DnaBuffer buf(DnaSeq::bytesneeded(a.size()) + DnaSeq::bytesneeded(b.size()));
    buf.push_back(a.data(), a.size());
    buf.push_back(b.data(), b.size());
    std::printf("%s", buf.getasciifilecontents().c_str());              // reference include/dnabuffer.hpp:33
    std::map<TKmer, int> ordered;                                         // std::less<Kmer<N>> (reference include/kmer.hpp:96-102)
    std::set<TKmer, std::less<TKmer>> s;
    std::unordered_map<TKmer, int> hashed;
    for (size_t i = 0; i < buf.size(); ++i)
        for (auto &k : TKmer::GetRepKmers(buf[i])) { ++ordered[k]; s.insert(k); ++hashed[k]; }
    TKmer prev; bool first = true, ok = ordered.size() == hashed.size() && s.size() == ordered.size();
    for (auto &kv : ordered) { if (!first && !(prev < kv.first)) ok = false; prev = kv.first; first = false; }
    std::printf("%zu %d\n", ordered.size(), ok ? 1 : 0);
    return ok ? 0 : 1;
}
'''


def test_shim_surface_compiles_for_clients_of_the_reference(tmp_path):
    """A client that uses DnaBuffer::getasciifilecontents and std::map<TKmer, ...> (std::less<Kmer<N>>) compiles against include/hysortk/."""
    (tmp_path / "surface.cpp").write_text(SURFACE_SRC)
    exe = str(tmp_path / "surface")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-DKMER_SIZE=31", "-DMINIMIZER_SIZE=17", "-DLOWER_KMER_FREQ=1",
                           "-DUPPER_KMER_FREQ=65535", "-DEXTENSION=0", "-o", exe, str(tmp_path / "surface.cpp"), "-L", os.path.join(util.ROOT, "hysortk_amd"), "-lhsk",
                           "-Wl,-rpath," + os.path.join(util.ROOT, "hysortk_amd")])
    out = subprocess.check_output([exe]).decode().splitlines()
    assert out[0] == "ACGTTGCAACGTACGTTTGACCATGACCAGTAGGATTACAGATTACA" and out[1] == "T" * 40 + "ACGT"
    n, ok = out[2].split()
    assert ok == "1" and int(n) == (47 - 31 + 1) + len({min(s, s.translate(str.maketrans("ACGT", "TGCA"))[::-1]) for s in [("T" * 40 + "ACGT")[i:i + 31] for i in range(14)]})


@pytest.mark.gpu
def test_example_driver_one_gbp_fasta_ingest_on_the_device(tmp_path):
    """read_dna_buffer() of the C++ shim on a 1 Gbp FASTA (1.03 GB of text, 60-base lines): the records are packed on the GPU
    (hsk_pack_fasta) and arrive in the DnaBuffer as one copy -- same histogram as the Python path on the same reads, same as the host
    packer (HSK_HOST_INGEST=1) on a tenth of the file, and the ingest itself well above what the per-base host loop reaches."""
    import re
    import numpy as np
    import hysortk_amd as H
    exe = str(tmp_path / "hysortk")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-DKMER_SIZE=31", "-DMINIMIZER_SIZE=17",
                           "-DLOWER_KMER_FREQ=2", "-DUPPER_KMER_FREQ=60", "-DEXTENSION=0", "-o", exe,
                           os.path.join(util.ROOT, "examples", "hysortk_main.cpp"), "-L", os.path.join(util.ROOT, "hysortk_amd"), "-lhsk",
                           "-Wl,-rpath," + os.path.join(util.ROOT, "hysortk_amd")])
    RL, LB = 150, 60
    rng = np.random.default_rng(12)
    genome = rng.integers(0, 4, 40_000_000, dtype=np.uint8)

    def write(path, nreads):
        starts = rng.integers(0, genome.size - RL, nreads)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        nl = (RL + LB - 1) // LB                                    # 3 lines: 60 + 60 + 30 bases
        w = 3 + RL + nl                                             # ">r\n" + bases + line breaks
        rec = np.empty((nreads, w), dtype=np.uint8)
        rec[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
        step = 1 << 19
        for a in range(0, nreads, step):
            st = starts[a:a + step]
            bases = lut[genome[st[:, None] + np.arange(RL)[None, :]]]
            o = 3
            for l in range(nl):
                n = min(LB, RL - l * LB)
                rec[a:a + st.size, o:o + n] = bases[:, l * LB:l * LB + n]; rec[a:a + st.size, o + n] = 10; o += n + 1
        with open(path, "wb") as f:
            f.write(rec.tobytes())
        with open(path + ".fai", "w") as f:
            for a in range(0, nreads, step):
                f.write("".join("r\t%d\t%d\t%d\t%d\n" % (RL, i * w + 3, LB, LB + 1) for i in range(a, min(a + step, nreads))))
        return os.path.getsize(path)
    big = str(tmp_path / "big.fa")
    nbytes = write(big, 1_000_000_000 // RL)
    so = subprocess.check_output([exe, big]).decode()
    m = re.search(r"read_dna_buffer: ([0-9.e+-]+) s, (\d+) reads, (\d+) packed bytes", so)
    assert m and int(m.group(2)) == 1_000_000_000 // RL and int(m.group(3)) == (1_000_000_000 // RL) * ((RL + 3) // 4)
    rate = nbytes / float(m.group(1)) / 1e9
    print("device ingest: %.2f GB/s of FASTA text (%.2f s for %.2f GB, GPU context created before)" % (rate, float(m.group(1)), nbytes / 1e9))
    hist_big = so[so.index("#count"):]
    # the same file through the Python mirror (device ingest + device-resident count): same histogram text
    dd = H.read_dna_buffer_device(H.Context(K=31, M=17, L=2, U=60), big)
    r = dd.count()
    assert H.histogram_text(r.histo) in so
    dd.free()
    assert rate >= 2.0, rate                                       # (measured 3.5 - 3.9 GB/s; the host loop: ~0.05 GB/s)
    # a smaller file through both ingest paths of the shim: identical output
    small = str(tmp_path / "small.fa")
    write(small, 800_000)                                           # 125 MB of text: above the 16 MB limit of the device path
    a = subprocess.check_output([exe, small]).decode()
    b = subprocess.check_output([exe, small], env=dict(os.environ, HSK_HOST_INGEST="1")).decode()
    assert a[a.index("#count"):] == b[b.index("#count"):] and "#count" in hist_big
