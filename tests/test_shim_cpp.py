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
