// ref_harness.cpp -- TEST INFRASTRUCTURE (runs only in the dev container, never on the GPU box).
//
// Our own driver around the REAL reference library objects built by oracle/build_ref.sh.  It
// includes the reference's headers from /root/reference at build time (nothing is copied into
// this repo) and dumps stage-level vectors as JSON so that tests/golden/make_golden.py can
// commit them as fixtures:
//   ref_harness stages <seqs.txt> <tot_tasks> [<tot_tasks> ...]   one ASCII read per line
//   ref_harness count  <fasta>                                    raw KmerListS via the public API
//   ref_harness murmur                                            known-answer keys
#include <mpi.h>
#include <omp.h>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include "hysortk.hpp"
#include "kmerops.hpp"
#include "supermer.hpp"
#include "hashfuncs.hpp"

using namespace hysortk;

static std::string hex(const uint8_t* p, size_t n)
{
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; ++i) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}
static std::string hex64(uint64_t v)
{
    char b[32]; snprintf(b, sizeof b, "\"%016llx\"", (unsigned long long)v); return b;
}

static int do_murmur()
{
    std::cout << "{\"murmur\": [";
    bool first = true;
    auto emit = [&](const std::vector<uint64_t>& key) {
        uint64_t h; murmurhash3_64(key.data(), (uint32_t)(8 * key.size()), &h);
        if (!first) std::cout << ", ";
        first = false;
        std::cout << "{\"key\": [";
        for (size_t i = 0; i < key.size(); ++i) std::cout << (i ? ", " : "") << hex64(key[i]);
        std::cout << "], \"hash\": " << hex64(h) << "}";
    };
    uint64_t x = 0x9e3779b97f4a7c15ULL;
    std::vector<uint64_t> fixed = {0, 1, 0x1be429f040000000ULL, 0x0123456789abcdefULL, ~0ULL};
    for (auto k : fixed) emit({k});
    for (int i = 0; i < 40; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; emit({x}); }
    for (int i = 0; i < 12; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint64_t y = x * 0xd1342543de82ef95ULL; emit({x, y}); }
    for (int i = 0; i < 12; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint64_t y = x * 0xd1342543de82ef95ULL; emit({x, y, x ^ y}); }
    std::cout << "]}" << std::endl;
    return 0;
}

static int do_stages(int argc, char** argv)
{
    std::ifstream in(argv[2]);
    std::vector<std::string> seqs;
    std::string line;
    while (std::getline(in, line)) seqs.push_back(line);
    std::vector<int> tasks;
    for (int i = 3; i < argc; ++i) tasks.push_back(atoi(argv[i]));

    size_t bufsize = 0;
    for (auto& s : seqs) bufsize += DnaSeq::bytesneeded(s.size());
    DnaBuffer buf(bufsize + 1);
    for (auto& s : seqs) buf.push_back(s.c_str(), s.size());

    std::cout << "{\"K\": " << KMER_SIZE << ", \"M\": " << MINIMIZER_SIZE << ", \"EXT\": " << EXTENSION << ", \"reads\": [";
    for (size_t r = 0; r < seqs.size(); ++r) {
        const DnaSeq& s = buf[r];
        if (r) std::cout << ", ";
        std::cout << "{\"seq\": \"" << seqs[r] << "\", \"packed\": \"" << hex(s.data(), s.numbytes()) << "\"";
        // canonical k-mers
        auto rep = TKmer::GetRepKmers(s);
        std::cout << ", \"repkmers\": [";
        for (size_t i = 0; i < rep.size(); ++i) {
            const uint64_t* w = (const uint64_t*)rep[i].GetBytes();
            std::cout << (i ? ", " : "") << "[";
            for (int j = 0; j < TKmer::NBYTES / 8; ++j) std::cout << (j ? ", " : "") << hex64(w[j]);
            std::cout << "]";
        }
        std::cout << "]";
        // canonical m-mer hashes
        auto mm = TMmer::GetRepMmers(s);
        std::cout << ", \"mmerhash\": [";
        for (size_t i = 0; i < mm.size(); ++i) std::cout << (i ? ", " : "") << hex64(mm[i].GetHash());
        std::cout << "]";
        // destinations + supermers for each task count
        std::cout << ", \"tasks\": {";
        for (size_t t = 0; t < tasks.size(); ++t) {
            int tot = tasks[t];
            DnaBuffer one(DnaSeq::bytesneeded(seqs[r].size()) + 1);
            one.push_back(seqs[r].c_str(), seqs[r].size());
            ParallelData data(1);
            FindKmerDestinationsParallel(one, 1, tot, data, 0);
            auto& dest = data.get_my_destinations(0)[0];
            std::cout << (t ? ", " : "") << "\"" << tot << "\": {\"dest\": [";
            for (size_t i = 0; i < dest.size(); ++i) std::cout << (i ? "," : "") << dest[i];
            std::cout << "], \"supermers\": [";
            std::vector<std::shared_ptr<ScatteredTask>> sca;
            for (int i = 0; i < tot; ++i) sca.push_back(std::make_shared<ScatteredSupermers>(i, 1));
            SupermerEncoder enc(sca, 0, MAX_SUPERMER_LEN);
            enc.encode(dest, one[0], 0);
            bool first = true;
            for (int i = 0; i < tot; ++i) {
                auto st = std::dynamic_pointer_cast<ScatteredSupermers>(sca[i]);
                auto& lens = st->get_length_buffer(0);
                auto& bytes = st->get_supermer_buffer(0);
                size_t off = 0;
                for (size_t j = 0; j < lens.size(); ++j) {
#if EXTENSION == 0
                    uint32_t len = lens[j]; uint32_t pos = 0;
#else
                    uint32_t len = lens[j].len; uint32_t pos = lens[j].pos;
#endif
                    size_t nb = cnt_bytes(len);
                    std::cout << (first ? "" : ", ") << "{\"task\": " << i << ", \"len\": " << len
                              << ", \"pos\": " << pos << ", \"bytes\": \"" << hex(bytes.data() + off, nb) << "\"}";
                    first = false;
                    off += nb;
                }
            }
            std::cout << "]}";
        }
        std::cout << "}}";
    }
    std::cout << "]}" << std::endl;
    return 0;
}

static int do_count(int argc, char** argv)
{
    std::string fasta = argv[2];
    auto dna = read_dna_buffer(fasta, MPI_COMM_WORLD);
    auto list = kmer_count(*dna, MPI_COMM_WORLD);
    // raw vector order (per-task ascending runs, kmerops.cpp:883-904), one entry per line:
    //   KMER \t cnt [\t pos,pos,... \t rid,rid,...]
    std::ofstream out(argv[3]);
    for (size_t i = 0; i < list->size(); ++i) {
        const auto& e = (*list)[i];
        out << e.kmer.GetString() << "\t" << e.cnt;
#if EXTENSION == 1
        out << "\t";
        for (size_t j = 0; j < e.pos.size(); ++j) out << (j ? "," : "") << e.pos[j];
        out << "\t";
        for (size_t j = 0; j < e.rid.size(); ++j) out << (j ? "," : "") << e.rid[j];
#endif
        out << "\n";
    }
    out.close();
    print_kmer_histogram(*list, MPI_COMM_WORLD);
    return 0;
}

int main(int argc, char** argv)
{
    MPI_Init(&argc, &argv);
    int rc = 1;
    if (argc >= 2 && std::string(argv[1]) == "murmur") rc = do_murmur();
    else if (argc >= 4 && std::string(argv[1]) == "stages") rc = do_stages(argc, argv);
    else if (argc >= 4 && std::string(argv[1]) == "count") rc = do_count(argc, argv);
    else std::cerr << "usage: ref_harness murmur | stages <seqs.txt> <tot_tasks>... | count <fasta> <out.txt>" << std::endl;
    MPI_Finalize();
    return rc;
}
