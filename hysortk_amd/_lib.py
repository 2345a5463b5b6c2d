"""ctypes binding of libhsk.so -- the C ABI declared in include/hsk.h.

The library is the only compute path: if it is missing this module raises (there is no Python /
CPU fallback), and hsk_init() fails with HSK_ERR_NO_DEVICE when no gfx950 GPU is present.
"""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))


class Config(C.Structure):
    _fields_ = [
        ("kmer_size", C.c_int32), ("minimizer_size", C.c_int32), ("lower_freq", C.c_int32), ("upper_freq", C.c_int32),
        ("extension", C.c_int32), ("ntasks", C.c_int32), ("device", C.c_int32), ("plain_dispatcher", C.c_int32),
        ("dispatch_upper_coe", C.c_double), ("dispatch_step", C.c_double),
        ("radix_bits", C.c_int32), ("flags", C.c_int32), ("unbalanced_ratio", C.c_double), ("tuning", C.c_char_p), ("reserved", C.c_int64 * 2),
    ]


class Result(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("nw", C.c_int32), ("ntasks", C.c_int32),
        ("entries", C.POINTER(C.c_uint64)), ("task_off", C.POINTER(C.c_uint64)),
        ("payload_off", C.POINTER(C.c_uint64)), ("pos", C.POINTER(C.c_uint32)), ("rid", C.POINTER(C.c_int32)),
        ("histo", C.POINTER(C.c_uint64)), ("histo_len", C.c_uint64), ("entries_dev", C.c_void_p),
        ("total_kmers", C.c_uint64), ("total_supermers", C.c_uint64), ("total_supermer_bytes", C.c_uint64),
        ("ms_total", C.c_double), ("ms_parse", C.c_double), ("ms_exchange", C.c_double), ("ms_extract", C.c_double),
        ("ms_sort", C.c_double), ("ms_count", C.c_double), ("ms_d2h", C.c_double), ("priv", C.c_void_p),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("scatter_launches", C.c_uint64), ("scatter_keys", C.c_uint64), ("scatter_bytes", C.c_uint64), ("scatter_ms", C.c_double),
        ("hist_launches", C.c_uint64), ("hist_bytes", C.c_uint64), ("hist_ms", C.c_double),
        ("fused_tasks", C.c_int64), ("redone_tasks", C.c_int64),
        ("agg_launches", C.c_uint64), ("agg_bytes", C.c_uint64), ("agg_ms", C.c_double),
        ("agg_retried_tasks", C.c_int64), ("parse_fallbacks", C.c_int64), ("heavy_tasks", C.c_int64), ("dropped_kmers", C.c_int64),
        ("scan_launches", C.c_uint64), ("scan_bytes", C.c_uint64), ("scan_ms", C.c_double),
        ("place_launches", C.c_uint64), ("place_supermers", C.c_uint64), ("place_ms", C.c_double),
        ("host_syncs", C.c_uint64), ("host_waits_covered", C.c_uint64), ("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
        ("bucket_launches", C.c_uint64), ("bucket_items", C.c_uint64), ("bucket_ms", C.c_double),
        ("combine_launches", C.c_uint64), ("combine_kmers", C.c_uint64), ("combine_pairs", C.c_uint64), ("combine_ms", C.c_double),
    ]


FLAG_PROFILE = 1
FLAG_KEEP_DEVICE = 2
FLAG_PLAIN_CLASSIFIER = 4
FLAG_NO_AGGREGATION = 8
FLAG_FULL_SORT = 16
FLAG_NO_COMBINE = 32
UNIQUE_ID_BYTES = 128

# every symbol include/hsk.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = [
    "hsk_abi_version", "hsk_device_count", "hsk_host_alloc", "hsk_host_free", "hsk_init", "hsk_destroy", "hsk_strerror", "hsk_last_error", "hsk_config_default",
    "hsk_count", "hsk_count_device", "hsk_count_loopback", "hsk_count_loopback_device", "hsk_result_free", "hsk_result_device_task", "hsk_format_entries", "hsk_get_stats",
    "hsk_stage_destinations", "hsk_stage_task_kmers", "hsk_stage_sort", "hsk_stage_count_sorted",
    "hsk_plan_tot_tasks", "hsk_plan_classify", "hsk_plan_dispatch", "hsk_plan_partition_reads", "hsk_plan_exchange",
    "hsk_comm_get_unique_id", "hsk_comm_init", "hsk_comm_destroy", "hsk_comm_selftest",
    "hsk_synth_reads", "hsk_synth_reads_err", "hsk_synth_free", "hsk_memcpy_d2h", "hsk_pack_fasta", "hsk_copy_peak",
]

_lib = None


def lib_path():
    # HSK_LIB: an alternative build of the same sources (kernel-variant experiments, tools/gpu_ab.sh)
    return os.environ.get("HSK_LIB") or os.path.join(_HERE, "libhsk.so")


def load():
    """Loads libhsk.so (building it first if the sources are newer).  Raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.environ.get("HSK_LIB") and (not os.path.exists(path) or _build.needs_build()):
        try:
            _build.build()
        except Exception as e:  # no hipcc on this machine: use the shipped .so if there is one
            if not os.path.exists(path):
                raise RuntimeError("libhsk.so is missing and could not be built: %s" % e)
    L = C.CDLL(path)
    vp, u64p = C.c_void_p, C.c_void_p
    L.hsk_abi_version.restype = C.c_int
    L.hsk_device_count.restype = C.c_int
    L.hsk_host_alloc.restype = C.c_void_p
    L.hsk_host_alloc.argtypes = [C.c_uint64]
    L.hsk_host_free.restype = None
    L.hsk_host_free.argtypes = [C.c_void_p]
    L.hsk_init.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.hsk_destroy.argtypes = [vp]
    L.hsk_destroy.restype = None
    L.hsk_strerror.restype = C.c_char_p
    L.hsk_strerror.argtypes = [C.c_int]
    L.hsk_last_error.restype = C.c_char_p
    L.hsk_last_error.argtypes = [vp]
    L.hsk_config_default.argtypes = [C.POINTER(Config)]
    L.hsk_config_default.restype = None
    L.hsk_count.argtypes = [vp, vp, C.c_uint64, u64p, vp, C.c_uint64, C.c_int64, C.POINTER(Result)]
    L.hsk_count_device.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_int64, C.POINTER(Result)]
    L.hsk_count_loopback.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(Result), vp, C.c_int32]
    L.hsk_count_loopback_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(Result), vp, C.c_int32]
    L.hsk_result_free.argtypes = [vp, C.POINTER(Result)]
    L.hsk_result_free.restype = None
    L.hsk_result_device_task.argtypes = [C.POINTER(Result), C.c_int32] + [vp] * 7
    L.hsk_get_stats.argtypes = [vp, C.POINTER(Stats), C.c_int]
    L.hsk_format_entries.argtypes = [vp, vp, C.c_uint64, C.c_int32, C.c_int32, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.hsk_stage_destinations.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, C.c_uint64, vp]
    L.hsk_stage_task_kmers.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_int64, C.c_int32, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.hsk_stage_sort.argtypes = [vp, vp, vp, C.c_uint64, C.c_int32]
    L.hsk_stage_count_sorted.argtypes = [vp, vp, C.c_uint64, C.c_int32, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.hsk_plan_tot_tasks.argtypes = [C.c_int] * 4
    L.hsk_plan_classify.argtypes = [vp, C.c_int, C.c_double, vp]
    L.hsk_plan_dispatch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]
    L.hsk_plan_partition_reads.argtypes = [vp, C.c_uint64, C.c_int, vp]
    L.hsk_plan_exchange.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.hsk_comm_get_unique_id.argtypes = [vp]
    L.hsk_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.hsk_comm_destroy.argtypes = [vp]
    L.hsk_comm_selftest.argtypes = [vp]
    L.hsk_synth_reads.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.hsk_synth_reads_err.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.hsk_synth_free.argtypes = [vp, vp, vp, vp]
    L.hsk_pack_fasta.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.hsk_memcpy_d2h.argtypes = [vp, vp, vp, C.c_uint64]
    L.hsk_copy_peak.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    _lib = L
    return L
