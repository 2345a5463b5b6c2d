// compiletime.h -- configuration contract of the drop-in headers.
// Same macro names and limits as the reference (its Makefile:1-46 passes them with -D, its
// include/compiletime.h:7-22 checks them); here they select template instances in the client and
// are forwarded to libhsk.so at run time through hsk_config.
#pragma once
#include <cstdint>
#include <limits>

#ifndef KMER_SIZE
#define KMER_SIZE 31
#endif
#ifndef MINIMIZER_SIZE
#define MINIMIZER_SIZE 17
#endif
#ifndef LOWER_KMER_FREQ
#define LOWER_KMER_FREQ 15
#endif
#ifndef UPPER_KMER_FREQ
#define UPPER_KMER_FREQ 40
#endif
#ifndef EXTENSION
#define EXTENSION 0
#endif

static_assert(KMER_SIZE > 2 && KMER_SIZE < 96, "KMER_SIZE must be in (2, 96)");
static_assert(KMER_SIZE % 32 != 0, "KMER_SIZE % 32 == 0 is undefined behaviour in the reference (kmer.hpp:260) and rejected here");
static_assert(MINIMIZER_SIZE > 0 && MINIMIZER_SIZE < KMER_SIZE && MINIMIZER_SIZE % 32 != 0,
              "MINIMIZER_SIZE must be < KMER_SIZE; MINIMIZER_SIZE % 32 == 0 is undefined behaviour in the reference (supermer.hpp:265) and rejected here");
static_assert(LOWER_KMER_FREQ > 0 && LOWER_KMER_FREQ <= UPPER_KMER_FREQ && UPPER_KMER_FREQ <= std::numeric_limits<uint16_t>::max(),
              "need 0 < LOWER_KMER_FREQ <= UPPER_KMER_FREQ <= 65535");
static_assert(EXTENSION == 0 || EXTENSION == 1, "EXTENSION is 0 or 1");

// MPI is optional: with -DHSK_WITH_MPI the real <mpi.h> is used and the four functions are
// collective over `comm` (one rank per GPU); without it MPI_Comm is a placeholder type so that
// client code written against the reference's signatures compiles unchanged for one process.
#ifdef HSK_WITH_MPI
#include <mpi.h>
#else
#ifndef HSK_MPI_STUB_DEFINED
#define HSK_MPI_STUB_DEFINED
typedef int MPI_Comm;
#ifndef MPI_COMM_WORLD
#define MPI_COMM_WORLD 0
#endif
#endif
#endif
