"""ctypes binding of the C++ host code (libbronko_host.so): index build and .bkdb codec (product code)."""
import ctypes as C
import os

import numpy as np

from .engine import BUCKET_INFO_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbronko_host.so")
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("bronko_amd: %s is missing -- build it with `make -C bronko_amd/host`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.bh_last_error.restype = C.c_char_p
    L.bh_index_build.restype = vp
    L.bh_index_build.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int]
    L.bh_index_build_mem.restype = vp
    L.bh_index_build_mem.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_int]
    L.bh_index_load.restype = vp
    L.bh_index_load.argtypes = [C.c_char_p]
    L.bh_index_save.restype = C.c_int
    L.bh_index_save.argtypes = [vp, C.c_char_p]
    L.bh_index_free.argtypes = [vp]
    for n, rt in (("bh_index_k", C.c_int), ("bh_index_meta_k", C.c_int), ("bh_index_n_buckets", C.c_uint64),
                  ("bh_index_n_entries", C.c_uint64), ("bh_index_bucket_ids", C.POINTER(C.c_uint64)),
                  ("bh_index_bucket_off", C.POINTER(C.c_uint64)), ("bh_index_entries", C.POINTER(C.c_uint8)),
                  ("bh_index_n_files", C.c_int), ("bh_index_total_cells", C.c_uint64)):
        getattr(L, n).restype = rt
        getattr(L, n).argtypes = [vp]
    L.bh_index_file_name.restype = C.c_char_p
    L.bh_index_file_name.argtypes = [vp, C.c_int]
    L.bh_index_n_seqs.restype = C.c_int
    L.bh_index_n_seqs.argtypes = [vp, C.c_int]
    L.bh_index_seq_name.restype = C.c_char_p
    L.bh_index_seq_name.argtypes = [vp, C.c_int, C.c_int]
    L.bh_index_seq_len.restype = C.c_uint64
    L.bh_index_seq_len.argtypes = [vp, C.c_int, C.c_int]
    L.bh_index_seq.restype = C.POINTER(C.c_uint8)
    L.bh_index_seq.argtypes = [vp, C.c_int, C.c_int]
    _lib = L
    return L


class HostIndex:
    """BronkoIndex (build.rs:23-28) held by the C++ host library."""

    def __init__(self, handle):
        L = load()
        if not handle:
            raise RuntimeError("bronko host: " + L.bh_last_error().decode(errors="replace"))
        self.h = C.c_void_p(handle)
        self.k = L.bh_index_k(self.h)
        self.meta_k = L.bh_index_meta_k(self.h)
        self.n_buckets = L.bh_index_n_buckets(self.h)
        self.n_entries = L.bh_index_n_entries(self.h)
        self.n_files = L.bh_index_n_files(self.h)
        self.total_cells = L.bh_index_total_cells(self.h)

    @classmethod
    def build(cls, k, fasta_paths, threads=4):
        arr = (C.c_char_p * len(fasta_paths))(*[p.encode() for p in fasta_paths])
        return cls(load().bh_index_build(k, arr, len(fasta_paths), threads))

    @classmethod
    def build_mem(cls, k, files, threads=4):
        """files: [(file_name, [(seq_name, seq_bytes), ...]), ...]"""
        fn = (C.c_char_p * len(files))(*[f[0].encode() for f in files])
        ns = (C.c_int * len(files))(*[len(f[1]) for f in files])
        flat = [s for f in files for s in f[1]]
        sn = (C.c_char_p * len(flat))(*[s[0].encode() for s in flat])
        sq = (C.c_char_p * len(flat))(*[bytes(s[1]) for s in flat])
        sl = (C.c_uint64 * len(flat))(*[len(s[1]) for s in flat])
        return cls(load().bh_index_build_mem(k, len(files), fn, ns, sn, sq, sl, threads))

    @classmethod
    def load(cls, path):
        return cls(load().bh_index_load(path.encode()))

    def save(self, path):
        L = load()
        if L.bh_index_save(self.h, path.encode()) != 0:
            raise RuntimeError("bronko host: " + L.bh_last_error().decode(errors="replace"))

    def close(self):
        if self.h:
            load().bh_index_free(self.h)
            self.h = None

    def bucket_ids(self):
        return np.ctypeslib.as_array(load().bh_index_bucket_ids(self.h), shape=(self.n_buckets,)).copy()

    def bucket_off(self):
        return np.ctypeslib.as_array(load().bh_index_bucket_off(self.h), shape=(self.n_buckets + 1,)).copy()

    def entries(self):
        raw = np.ctypeslib.as_array(load().bh_index_entries(self.h), shape=(max(self.n_entries, 1) * 12,))
        return raw[: self.n_entries * 12].copy().view(BUCKET_INFO_DTYPE)

    def files(self):
        L = load()
        out = []
        for f in range(self.n_files):
            seqs = []
            for s in range(L.bh_index_n_seqs(self.h, f)):
                n = L.bh_index_seq_len(self.h, f, s)
                seq = bytes(np.ctypeslib.as_array(L.bh_index_seq(self.h, f, s), shape=(n,))) if n else b""
                seqs.append((L.bh_index_seq_name(self.h, f, s).decode(), seq))
            out.append((L.bh_index_file_name(self.h, f).decode(), seqs))
        return out

    def genome_len(self, f):
        L = load()
        return sum(L.bh_index_seq_len(self.h, f, s) for s in range(L.bh_index_n_seqs(self.h, f)))

    def engine(self, params=None):
        """bk_engine_create on this index."""
        from .engine import Engine
        return Engine(self.k, self.bucket_ids(), self.bucket_off(), self.entries(), self.files(), params)


class CallParams(C.Structure):  # bh_call_params == bronko::CallParams (cli.rs:92-135 defaults from consts.rs)
    _fields_ = [("k", C.c_int32), ("min_af", C.c_double), ("no_end_filter", C.c_int32), ("no_strand_filter", C.c_int32),
                ("no_strand_balance_filter", C.c_int32), ("strand_balance_ratio", C.c_double), ("n_per_strand", C.c_uint64),
                ("strand_odds_max", C.c_double), ("min_depth", C.c_uint64), ("min_variant_depth", C.c_uint64),
                ("variant_multiplier", C.c_double)]


def default_call_params(k=21):
    return CallParams(k, 0.03, 0, 0, 0, 0.1, 2, 6.0, 300, 3, 1.5)


def _caller_lib():
    L = load()
    if not hasattr(L, "_caller_ready"):
        vp = C.c_void_p
        L.bh_pick_best_genome.restype = C.c_int
        L.bh_pick_best_genome.argtypes = [vp, vp, vp]
        L.bh_baseline_noise_max.restype = None
        L.bh_baseline_noise_max.argtypes = [vp, vp, C.c_uint64, vp]
        L.bh_call_and_write.restype = C.c_int
        L.bh_call_and_write.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.POINTER(CallParams), C.c_char_p, C.c_char_p, C.c_char_p, vp, vp]
        L.bh_clean_sample_id.restype = None
        L.bh_clean_sample_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L._caller_ready = True
    return L


def pick_best_genome(ix, stats, present):
    stats = np.ascontiguousarray(stats, np.uint64)
    present = np.ascontiguousarray(present, np.uint8)
    return _caller_lib().bh_pick_best_genome(ix.h, stats.ctypes.data, present.ctypes.data)


def baseline_noise_max(fwd4, rev4):
    fwd4 = np.ascontiguousarray(fwd4, np.uint64)
    rev4 = np.ascontiguousarray(rev4, np.uint64)
    out = np.zeros(len(fwd4) // 4)
    _caller_lib().bh_baseline_noise_max(fwd4.ctypes.data, rev4.ctypes.data, len(out), out.ctypes.data)
    return out


def call_and_write(ix, file_id, arrays, params, vcf_path=None, reads_path="", pileup_path=None):
    """call_variants + writers on four pileup arrays.  Returns (n_records, n_major, n_minor, breadth, depth)."""
    a = [np.ascontiguousarray(x, np.uint64) for x in arrays]
    summary = np.zeros(3, np.uint64)
    cov = np.zeros(2)
    L = _caller_lib()
    rc = L.bh_call_and_write(ix.h, file_id, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                             C.byref(params), vcf_path.encode() if vcf_path else None, reads_path.encode(),
                             pileup_path.encode() if pileup_path else None, summary.ctypes.data, cov.ctypes.data)
    if rc != 0:
        raise RuntimeError("bronko host: " + L.bh_last_error().decode(errors="replace"))
    return int(summary[0]), int(summary[1]), int(summary[2]), float(cov[0]), float(cov[1])


def clean_sample_id(path):
    buf = C.create_string_buffer(4096)
    _caller_lib().bh_clean_sample_id(path.encode(), buf, 4096)
    return buf.value.decode()
