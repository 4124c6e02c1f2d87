"""ctypes binding of include/bronko_hip.h (libbronko_hip.so).  No fallbacks: a missing library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BRONKO_HIP_LIB: another build of the same library, for A/B timing of two builds in one run -- tools/ab.sh; the harness reads it, the
# library itself reads no environment)
LIB_PATH = os.environ.get("BRONKO_HIP_LIB") or os.path.join(_HERE, "libbronko_hip.so")
# the -DBK_TESTING build of the same sources: BK_* environment variables that force a code path exist only there
TESTING_LIB_PATH = os.environ.get("BRONKO_HIP_TESTING_LIB") or os.path.join(_HERE, "libbronko_hip_testing.so")


class BucketInfo(C.Structure):  # include/bronko_hip.h bk_bucket_info == build.rs:52-60
    _fields_ = [("file_id", C.c_uint16), ("seq_id", C.c_uint8), ("location", C.c_uint32),
                ("idx", C.c_uint8), ("canonical", C.c_uint8)]


class IndexDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("n_buckets", C.c_uint64), ("bucket_ids", C.c_void_p), ("bucket_off", C.c_void_p),
                ("entries", C.c_void_p), ("n_entries", C.c_uint64), ("n_files", C.c_int32), ("n_seqs", C.c_void_p),
                ("seq_lens", C.c_void_p), ("seqs", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("n_fixed", C.c_int32), ("use_full_kmer", C.c_int32), ("ci", C.c_uint64), ("cs", C.c_uint64),
                ("cx", C.c_uint64), ("device", C.c_int32), ("full_kmer_stats", C.c_int32),
                ("kmer_table_log2", C.c_uint32), ("pileup_selected_only", C.c_uint32)]


class CallParams(C.Structure):  # include/bronko_hip.h bk_call_params
    _fields_ = [("k", C.c_int32), ("no_end_filter", C.c_int32), ("no_strand_filter", C.c_int32), ("no_strand_balance_filter", C.c_int32),
                ("min_af", C.c_double), ("strand_balance_ratio", C.c_double), ("strand_odds_max", C.c_double),
                ("variant_multiplier", C.c_double), ("n_per_strand", C.c_uint64), ("min_depth", C.c_uint64),
                ("min_variant_depth", C.c_uint64)]


class CallRecord(C.Structure):  # bk_call_record
    _fields_ = [("seq_id", C.c_int32), ("ref_base", C.c_uint8), ("alt_base", C.c_uint8), ("pad", C.c_uint16), ("pos", C.c_uint64),
                ("fwd_ref", C.c_uint64), ("rev_ref", C.c_uint64), ("fwd_alt", C.c_uint64), ("rev_alt", C.c_uint64),
                ("depth", C.c_uint64), ("af", C.c_double), ("sor", C.c_double)]


class CallSummary(C.Structure):  # bk_call_summary
    _fields_ = [("file_id", C.c_int32), ("pad", C.c_uint32), ("n_records", C.c_uint64), ("n_major", C.c_uint64), ("n_minor", C.c_uint64),
                ("covered", C.c_uint64), ("positions", C.c_uint64), ("coverage", C.c_uint64)]


class BuiltIndex(C.Structure):  # bk_built_index
    _fields_ = [("n_buckets", C.c_uint64), ("n_entries", C.c_uint64), ("bucket_ids", C.c_void_p), ("bucket_off", C.c_void_p),
                ("entries", C.c_void_p)]


# every symbol include/bronko_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = ["bk_abi_version", "bk_device_count", "bk_device_memory", "bk_last_error", "bk_params_default", "bk_engine_create", "bk_engine_destroy", "bk_engine_fork", "bk_engine_fork_params", "bk_engine_get_stream",
           "bk_engine_set_stream", "bk_total_cells", "bk_n_files", "bk_n_slots", "bk_counter_len", "bk_can_shard", "bk_sample_begin",
           "bk_push_reads_packed", "bk_push_reads_packed_device", "bk_push_reads_ascii", "bk_push_reads_ascii_device", "bk_counters_device_ptr", "bk_sample_finalize",
           "bk_sample_finalize_shard", "bk_shard_measure", "bk_shard_transport", "bk_shard_received", "bk_transport_overflow", "bk_shard_sums_device_ptr", "bk_sample_merge_shards", "bk_kmer_table_partition", "bk_kmer_table_replace",
           "bk_pileup_device_ptr", "bk_sample_download", "bk_sample_finish", "bk_pack_reads", "bk_pack_reads_flat",
           "bk_timing_enable", "bk_timing_read", "bk_call_params_default", "bk_sample_call", "bk_sample_download_calls", "bk_sample_download_noise",
           "bk_build_index", "bk_built_index_free", "bk_build_last_error"]

_libs = {}
_testing = False


def use_testing_library(flag=True):
    """Engines created from now on bind libbronko_hip_testing.so (tests that force a path through a BK_* variable, profiling
    tools); the product and bench.py never call this."""
    global _testing
    _testing = bool(flag)


def load(testing=None):
    testing = _testing if testing is None else bool(testing)
    if testing in _libs:
        return _libs[testing]
    path = TESTING_LIB_PATH if testing else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError("bronko_amd: %s is missing -- build it with `make -C bronko_amd/csrc` "
                           "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    L.bk_abi_version.restype = C.c_int
    L.bk_device_count.restype = C.c_int
    L.bk_device_memory.restype = C.c_int
    L.bk_device_memory.argtypes = [C.c_int, C.POINTER(u64), C.POINTER(u64)]
    L.bk_last_error.restype = C.c_char_p
    L.bk_params_default.argtypes = [C.POINTER(Params)]
    L.bk_engine_create.restype = C.c_int
    L.bk_engine_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(Params), C.POINTER(vp)]
    L.bk_engine_destroy.argtypes = [vp]
    L.bk_engine_get_stream.restype = vp
    L.bk_engine_get_stream.argtypes = [vp]
    L.bk_engine_fork.restype = C.c_int
    L.bk_engine_fork.argtypes = [vp, C.POINTER(vp)]
    L.bk_engine_fork_params.restype = C.c_int
    L.bk_engine_fork_params.argtypes = [vp, C.POINTER(Params), C.POINTER(vp)]
    L.bk_push_reads_ascii_device.restype = C.c_int
    L.bk_push_reads_ascii_device.argtypes = [vp, C.c_int, vp, vp, u64, u64, u32]
    L.bk_engine_set_stream.restype = C.c_int
    L.bk_engine_set_stream.argtypes = [vp, vp]
    for n, rt in (("bk_total_cells", u64), ("bk_n_files", i32), ("bk_n_slots", u64), ("bk_counter_len", u64)):
        getattr(L, n).restype = rt
        getattr(L, n).argtypes = [vp]
    L.bk_sample_begin.restype = C.c_int
    L.bk_sample_begin.argtypes = [vp]
    L.bk_push_reads_packed.restype = C.c_int
    L.bk_push_reads_packed.argtypes = [vp, C.c_int, vp, u32, vp, u64]
    L.bk_push_reads_packed_device.restype = C.c_int
    L.bk_push_reads_packed_device.argtypes = [vp, C.c_int, vp, u32, vp, u64]
    L.bk_push_reads_ascii.restype = C.c_int
    L.bk_push_reads_ascii.argtypes = [vp, C.c_int, vp, vp, u64]
    L.bk_counters_device_ptr.restype = C.c_int
    L.bk_counters_device_ptr.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.bk_sample_finalize.restype = C.c_int
    L.bk_sample_finalize.argtypes = [vp, C.c_int]
    L.bk_sample_finalize_shard.restype = C.c_int
    L.bk_sample_finalize_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.bk_shard_measure.restype = C.c_int
    L.bk_shard_measure.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.bk_shard_transport.restype = C.c_int
    L.bk_shard_transport.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp)]
    L.bk_shard_received.restype = C.c_int
    L.bk_shard_received.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.bk_transport_overflow.restype = C.c_int
    L.bk_transport_overflow.argtypes = [vp, C.POINTER(C.c_int)]
    L.bk_kmer_table_partition.restype = C.c_int
    L.bk_kmer_table_partition.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.bk_kmer_table_replace.restype = C.c_int
    L.bk_kmer_table_replace.argtypes = [vp, vp, vp, u64]
    L.bk_shard_sums_device_ptr.restype = C.c_int
    L.bk_shard_sums_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.bk_sample_merge_shards.restype = C.c_int
    L.bk_sample_merge_shards.argtypes = [vp]
    L.bk_pileup_device_ptr.restype = C.c_int
    L.bk_pileup_device_ptr.argtypes = [vp, C.POINTER(vp)]
    L.bk_sample_download.restype = C.c_int
    L.bk_sample_download.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.bk_sample_finish.restype = C.c_int
    L.bk_sample_finish.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.bk_pack_reads.restype = u64
    L.bk_pack_reads.argtypes = [vp, vp, u64, i32, u32, vp, vp, u64]
    L.bk_pack_reads_flat.restype = u64
    L.bk_pack_reads_flat.argtypes = [vp, vp, u64, i32, u32, vp, vp, u64]
    L.bk_call_params_default.argtypes = [C.POINTER(CallParams)]
    L.bk_sample_call.restype = C.c_int
    L.bk_sample_call.argtypes = [vp, C.c_int, C.POINTER(CallParams)]
    L.bk_sample_download_calls.restype = C.c_int
    L.bk_sample_download_calls.argtypes = [vp, C.POINTER(CallSummary), vp, u64]
    L.bk_sample_download_noise.restype = C.c_int
    L.bk_sample_download_noise.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.bk_build_index.restype = C.c_int
    L.bk_build_index.argtypes = [i32, i32, vp, vp, vp, i32, C.POINTER(BuiltIndex)]
    L.bk_built_index_free.argtypes = [C.POINTER(BuiltIndex)]
    L.bk_build_last_error.restype = C.c_char_p
    L.bk_timing_enable.restype = C.c_int
    L.bk_timing_enable.argtypes = [vp, C.c_int]
    L.bk_timing_read.restype = C.c_int
    L.bk_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), C.c_int]
    _libs[testing] = L
    return L
