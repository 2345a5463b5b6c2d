"""hysortk_amd -- MI355X-native sort-based k-mer counting behind the HySortK library surface.

The hot path (supermer/k-mer extraction, per-task LSD radix sort, adjacent-equal merge-count,
RCCL supermer exchange) lives in libhsk.so (csrc/, C ABI in include/hsk.h); this package is the
Python host mirror of the reference's four public functions and types.
"""
from .api import (Context, DnaBuffer, DnaSeq, HskError, KmerList, histogram_text, kmer_count, pack_sequence,  # noqa: F401
                  plan_classify, plan_dispatch, plan_exchange, plan_partition_reads, plan_tot_tasks, print_kmer_histogram,
                  read_dna_buffer, read_fai, write_output_file, paradis_order, DeviceDna, DeviceResult, read_dna_buffer_device, pinned_empty, pinned_free)

__version__ = "0.1.0"
