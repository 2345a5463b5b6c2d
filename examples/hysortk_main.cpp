// hysortk_main.cpp -- the reference's standalone driver (standalone/main.cpp:9-72) written
// against the drop-in headers: `hysortk <fasta> [outdir]` (needs <fasta>.fai next to the FASTA).
// Build: g++ -O2 -std=c++17 -Iinclude -DKMER_SIZE=31 -DMINIMIZER_SIZE=17 -DLOWER_KMER_FREQ=15
//        -DUPPER_KMER_FREQ=40 -DEXTENSION=0 examples/hysortk_main.cpp -Lhysortk_amd -lhsk
#include <chrono>
#include <iostream>
#include "hysortk/hysortk.hpp"

int main(int argc, char **argv)
{
#ifdef HSK_WITH_MPI
    MPI_Init(&argc, &argv);
#endif
    if (argc < 2) { std::cerr << "Usage: " << argv[0] << " <fasta file> <output dir>(Optional)" << std::endl; return 1; }
    const std::string fasta = argv[1];
    std::cout << "KMER_SIZE: " << KMER_SIZE << " MINIMIZER_SIZE: " << MINIMIZER_SIZE << " LOWER_KMER_FREQ: " << LOWER_KMER_FREQ
              << " UPPER_KMER_FREQ: " << UPPER_KMER_FREQ << " EXTENSION: " << EXTENSION << std::endl;
    try {
        (void)hysortk::detail::context(MPI_COMM_WORLD);                 // the process's GPU context first (HIP runtime + code objects: not ingest time)
        const auto tr = std::chrono::steady_clock::now();
        auto dna = hysortk::read_dna_buffer(fasta, MPI_COMM_WORLD);
        const double sr = std::chrono::duration<double>(std::chrono::steady_clock::now() - tr).count();
        std::cout << "read_dna_buffer: " << sr << " s, " << dna->size() << " reads, " << dna->getbufsize() << " packed bytes" << std::endl;
        const auto t0 = std::chrono::steady_clock::now();
        auto list = hysortk::kmer_count(*dna, MPI_COMM_WORLD);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "Overall kmer counting (Excluding I/O): " << s << " s, " << list->size() << " k-mers kept" << std::endl;
        hysortk::print_kmer_histogram(*list, MPI_COMM_WORLD);
        if (argc >= 3) hysortk::write_output_file(*list, argv[2], MPI_COMM_WORLD);
    } catch (const std::exception &e) {
        std::cerr << "hysortk: " << e.what() << std::endl;
        return 1;
    }
#ifdef HSK_WITH_MPI
    MPI_Finalize();
#endif
    return 0;
}
