"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (bronko_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("bronko_oracle.c", "bronko_oracle_mt.c", "bronko_oracle.h", "tcrit_table.inc")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


class BucketInfo(C.Structure):  # build.rs:52-60 #[repr(C)]
    _fields_ = [("file_id", C.c_uint16), ("seq_id", C.c_uint8), ("location", C.c_uint32),
                ("idx", C.c_uint8), ("canonical", C.c_uint8)]


BUCKET_INFO_DTYPE = np.dtype({"names": ["file_id", "seq_id", "location", "idx", "canonical"],
                              "formats": [np.uint16, np.uint8, np.uint32, np.uint8, np.uint8],
                              "offsets": [0, 2, 4, 8, 9], "itemsize": 12})


class CallParams(C.Structure):
    _fields_ = [("k", C.c_int), ("min_af", C.c_double), ("no_end_filter", C.c_int), ("no_strand_filter", C.c_int),
                ("no_strand_balance_filter", C.c_int), ("strand_balance_ratio", C.c_double),
                ("n_per_strand", C.c_uint64), ("strand_odds_max", C.c_double), ("min_depth", C.c_uint64),
                ("min_variant_depth", C.c_uint64), ("variant_multiplier", C.c_double)]


class MapParams(C.Structure):
    _fields_ = [("n_fixed", C.c_int), ("use_full_kmer", C.c_int), ("ci", C.c_uint64), ("cs", C.c_uint64),
                ("cx", C.c_uint64)]


class VcfRecord(C.Structure):
    _fields_ = [("seq_id", C.c_int), ("pos", C.c_uint64), ("ref_base", C.c_uint8), ("alt_base", C.c_uint8),
                ("fwd_ref", C.c_uint64), ("rev_ref", C.c_uint64), ("fwd_alt", C.c_uint64), ("rev_alt", C.c_uint64),
                ("depth", C.c_uint64), ("af", C.c_double), ("sor", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u64p = C.POINTER(C.c_uint64)
    u8p = C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.orc_last_error.restype = C.c_char_p
    L.orc_nt_to_bits.restype = C.c_uint8
    L.orc_nt_to_bits.argtypes = [C.c_uint8]
    L.orc_kmer_to_u64.restype = C.c_uint64
    L.orc_kmer_to_u64.argtypes = [C.c_char_p, C.c_int]
    L.orc_reverse_complement_u64.restype = C.c_uint64
    L.orc_reverse_complement_u64.argtypes = [C.c_uint64, C.c_int]
    L.orc_canonical_kmer.restype = C.c_uint64
    L.orc_canonical_kmer.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.orc_assign_buckets.restype = None
    L.orc_assign_buckets.argtypes = [C.c_uint64, C.c_int, u64p]
    L.orc_index_build.restype = vp
    L.orc_index_build.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int]
    L.orc_index_build_mem.restype = vp
    L.orc_index_build_mem.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p]
    L.orc_bkdb_load.restype = vp
    L.orc_bkdb_load.argtypes = [C.c_char_p]
    L.orc_bkdb_save.restype = C.c_int
    L.orc_bkdb_save.argtypes = [vp, C.c_char_p]
    L.orc_index_free.argtypes = [vp]
    for name, rt in (("orc_index_k", C.c_int), ("orc_index_meta_k", C.c_int), ("orc_index_n_buckets", C.c_uint64),
                     ("orc_index_n_entries", C.c_uint64), ("orc_index_bucket_ids", u64p),
                     ("orc_index_bucket_off", u64p), ("orc_index_entries", C.POINTER(BucketInfo)),
                     ("orc_index_n_files", C.c_int), ("orc_index_total_cells", C.c_uint64)):
        f = getattr(L, name)
        f.restype = rt
        f.argtypes = [vp]
    L.orc_index_file_name.restype = C.c_char_p
    L.orc_index_file_name.argtypes = [vp, C.c_int]
    L.orc_index_n_seqs.restype = C.c_int
    L.orc_index_n_seqs.argtypes = [vp, C.c_int]
    L.orc_index_seq_name.restype = C.c_char_p
    L.orc_index_seq_name.argtypes = [vp, C.c_int, C.c_int]
    L.orc_index_seq_len.restype = C.c_uint64
    L.orc_index_seq_len.argtypes = [vp, C.c_int, C.c_int]
    L.orc_index_seq.restype = u8p
    L.orc_index_seq.argtypes = [vp, C.c_int, C.c_int]
    L.orc_index_cell_offset.restype = C.c_uint64
    L.orc_index_cell_offset.argtypes = [vp, C.c_int, C.c_int]
    L.orc_index_lookup.restype = C.c_uint64
    L.orc_index_lookup.argtypes = [vp, C.c_uint64, u64p]
    L.orc_counter_new.restype = vp
    L.orc_counter_new.argtypes = [C.c_int]
    L.orc_counter_add_read.argtypes = [vp, C.c_char_p, C.c_uint64]
    L.orc_counter_add_fastq.restype = C.c_uint64
    L.orc_counter_add_fastq.argtypes = [vp, C.c_char_p]
    L.orc_counter_finish.restype = C.c_uint64
    L.orc_counter_finish.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, u64p]
    L.orc_counter_kmers.restype = u64p
    L.orc_counter_kmers.argtypes = [vp]
    L.orc_counter_counts.restype = u64p
    L.orc_counter_counts.argtypes = [vp]
    L.orc_counter_free.argtypes = [vp]
    L.orc_map_kmers.restype = None
    L.orc_map_kmers.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orc_pick_best_genome.restype = C.c_int
    L.orc_pick_best_genome.argtypes = [vp, vp, vp]
    L.orc_call_params_default.argtypes = [C.POINTER(CallParams)]
    L.orc_map_params_default.argtypes = [C.POINTER(MapParams)]
    L.orc_baseline_noise.restype = None
    L.orc_baseline_noise.argtypes = [vp, vp, C.c_uint64, vp, vp, vp]
    L.orc_call_variants.restype = C.c_uint64
    L.orc_call_variants.argtypes = [vp, C.c_int, C.POINTER(CallParams), vp, vp, vp, vp,
                                    C.POINTER(C.POINTER(VcfRecord)), u64p, u64p, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]
    L.orc_free.argtypes = [vp]
    L.orc_clean_sample_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.orc_write_vcf.restype = C.c_int
    L.orc_write_vcf.argtypes = [C.c_char_p, C.c_char_p, vp, C.c_int, C.POINTER(VcfRecord), C.c_uint64]
    L.orc_write_pileup.restype = C.c_int
    L.orc_write_pileup.argtypes = [C.c_char_p, vp, C.c_int, vp, vp]
    L.orc_sample_pileup_mt.restype = None
    L.orc_sample_pileup_mt.argtypes = [vp, C.POINTER(MapParams), C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp,
                                       vp, C.POINTER(C.c_double)]
    L.orc_sample_pileup.restype = None
    L.orc_sample_pileup.argtypes = [vp, C.POINTER(MapParams), C.c_int, C.POINTER(C.c_char_p), u64p, u64p,
                                    vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


# ------------------------------------------------------------------ thin pythonic helpers

def assign_buckets(kmer, k):
    out = (C.c_uint64 * k)()
    lib().orc_assign_buckets(C.c_uint64(kmer), k, out)
    return list(out)


def kmer_to_u64(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    return lib().orc_kmer_to_u64(b, len(b))


def reverse_complement_u64(v, k):
    return lib().orc_reverse_complement_u64(C.c_uint64(v), k)


def canonical_kmer(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    rc = C.c_int(0)
    v = lib().orc_canonical_kmer(b, len(b), C.byref(rc))
    return v, bool(rc.value)


def clean_sample_id(path):
    buf = C.create_string_buffer(4096)
    lib().orc_clean_sample_id(path.encode(), buf, 4096)
    return buf.value.decode()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Index:
    """Decoded / built BronkoIndex (build.rs:23-28) held by the oracle library."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        self.h = C.c_void_p(handle)
        L = lib()
        self.k = L.orc_index_k(self.h)
        self.meta_k = L.orc_index_meta_k(self.h)
        self.n_buckets = L.orc_index_n_buckets(self.h)
        self.n_entries = L.orc_index_n_entries(self.h)
        self.n_files = L.orc_index_n_files(self.h)
        self.total_cells = L.orc_index_total_cells(self.h)

    @classmethod
    def build(cls, k, fasta_paths):
        arr = (C.c_char_p * len(fasta_paths))(*[p.encode() for p in fasta_paths])
        return cls(lib().orc_index_build(k, arr, len(fasta_paths)))

    @classmethod
    def build_mem(cls, k, files):
        """files: list of (file_name, [(seq_name, bytes), ...])"""
        fn = (C.c_char_p * len(files))(*[f[0].encode() for f in files])
        ns = (C.c_int * len(files))(*[len(f[1]) for f in files])
        flat = [s for f in files for s in f[1]]
        sn = (C.c_char_p * len(flat))(*[s[0].encode() for s in flat])
        sq = (C.c_char_p * len(flat))(*[bytes(s[1]) for s in flat])
        sl = (C.c_uint64 * len(flat))(*[len(s[1]) for s in flat])
        return cls(lib().orc_index_build_mem(k, len(files), fn, ns, sn, sq, sl))

    @classmethod
    def load(cls, path):
        return cls(lib().orc_bkdb_load(path.encode()))

    def save(self, path):
        if lib().orc_bkdb_save(self.h, path.encode()) != 0:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())

    def close(self):
        if self.h:
            lib().orc_index_free(self.h)
            self.h = None

    # flattened views (copies)
    def bucket_ids(self):
        return np.ctypeslib.as_array(lib().orc_index_bucket_ids(self.h), shape=(self.n_buckets,)).copy()

    def bucket_off(self):
        return np.ctypeslib.as_array(lib().orc_index_bucket_off(self.h), shape=(self.n_buckets + 1,)).copy()

    def entries(self):
        p = lib().orc_index_entries(self.h)
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.n_entries * 12,)).copy()
        # padding bytes (offsets 3, 10, 11) are zeroed by the oracle
        return raw.view(BUCKET_INFO_DTYPE)

    def files(self):
        """[(file_name, [(seq_name, seq_bytes), ...]), ...]"""
        L = lib()
        out = []
        for f in range(self.n_files):
            seqs = []
            for s in range(L.orc_index_n_seqs(self.h, f)):
                n = L.orc_index_seq_len(self.h, f, s)
                seq = bytes(np.ctypeslib.as_array(L.orc_index_seq(self.h, f, s), shape=(n,))) if n else b""
                seqs.append((L.orc_index_seq_name(self.h, f, s).decode(), seq))
            out.append((L.orc_index_file_name(self.h, f).decode(), seqs))
        return out

    def cell_offset(self, f, s):
        return lib().orc_index_cell_offset(self.h, f, s)

    def genome_cells(self, f):
        """(first_cell, n_cells) of file f in the flat pileup arrays"""
        L = lib()
        ns = L.orc_index_n_seqs(self.h, f)
        first = L.orc_index_cell_offset(self.h, f, 0) if ns else 0
        n = sum(L.orc_index_seq_len(self.h, f, s) for s in range(ns))
        return first, n

    def sequence_cells(self, f):
        """[(first_cell, length)] of the sequences of file f"""
        L = lib()
        return [(L.orc_index_cell_offset(self.h, f, s), L.orc_index_seq_len(self.h, f, s)) for s in range(L.orc_index_n_seqs(self.h, f))]


def count_kmers(k, reads, ci=3, cs=1000000, cx=1000000000):
    """KMC contract on a list of ASCII reads -> (kmers u64[], counts u64[], stats[4])"""
    L = lib()
    c = C.c_void_p(L.orc_counter_new(k))
    for r in reads:
        L.orc_counter_add_read(c, bytes(r), len(r))
    stats = (C.c_uint64 * 4)()
    n = L.orc_counter_finish(c, ci, cs, cx, stats)
    km = np.ctypeslib.as_array(L.orc_counter_kmers(c), shape=(max(n, 1),))[:n].copy()
    ct = np.ctypeslib.as_array(L.orc_counter_counts(c), shape=(max(n, 1),))[:n].copy()
    L.orc_counter_free(c)
    return km, ct, list(stats)


def count_kmers_fastq(k, path, ci=3, cs=1000000, cx=1000000000):
    L = lib()
    c = C.c_void_p(L.orc_counter_new(k))
    nr = L.orc_counter_add_fastq(c, path.encode())
    if nr == 2 ** 64 - 1:
        raise RuntimeError("oracle: " + L.orc_last_error().decode())
    stats = (C.c_uint64 * 4)()
    n = L.orc_counter_finish(c, ci, cs, cx, stats)
    km = np.ctypeslib.as_array(L.orc_counter_kmers(c), shape=(max(n, 1),))[:n].copy()
    ct = np.ctypeslib.as_array(L.orc_counter_counts(c), shape=(max(n, 1),))[:n].copy()
    L.orc_counter_free(c)
    return km, ct, list(stats)


class Pileup:
    """The four (genome, seq, pos, base) arrays of initialize_output_maps (call.rs:1437-1480) + stats."""

    def __init__(self, ix, n_mates=1):
        n = ix.total_cells * 4
        self.fwd_depth = np.zeros(n, np.uint64)
        self.rev_depth = np.zeros(n, np.uint64)
        self.fwd_nk = np.zeros(n, np.uint64)
        self.rev_nk = np.zeros(n, np.uint64)
        self.stats = np.zeros((n_mates, ix.n_files, 3), np.uint64)
        self.present = np.zeros((n_mates, ix.n_files), np.uint8)
        self.kmc_stats = np.zeros((n_mates, 4), np.uint64)

    def arrays(self):
        return self.fwd_depth, self.rev_depth, self.fwd_nk, self.rev_nk


def map_kmers(ix, kmers, counts, pile, mate=0, n_fixed=2, use_full_kmer=False):
    kmers = np.ascontiguousarray(kmers, np.uint64)
    counts = np.ascontiguousarray(counts, np.uint64)
    lib().orc_map_kmers(ix.h, _ptr(kmers), _ptr(counts), len(kmers), n_fixed, int(use_full_kmer),
                        _ptr(pile.fwd_depth), _ptr(pile.rev_depth), _ptr(pile.fwd_nk), _ptr(pile.rev_nk),
                        _ptr(pile.stats[mate]), _ptr(pile.present[mate]))


def sample_pileup(ix, mates, n_fixed=2, use_full_kmer=False, ci=3, cs=1000000, cx=1000000000):
    """mates: list (1 = single-end, 2 = paired) of lists of ASCII reads.  Returns a filled Pileup."""
    pile = Pileup(ix, len(mates))
    mp = MapParams(n_fixed, int(use_full_kmer), ci, cs, cx)
    flat = [bytes(r) for m in mates for r in m]
    arr = (C.c_char_p * max(len(flat), 1))(*flat)
    lens = np.array([len(r) for r in flat] + [0], np.uint64)
    off = np.zeros(len(mates) + 1, np.uint64)
    off[1:] = np.cumsum([len(m) for m in mates])
    lib().orc_sample_pileup(ix.h, C.byref(mp), len(mates), arr, lens.ctypes.data_as(C.POINTER(C.c_uint64)),
                            off.ctypes.data_as(C.POINTER(C.c_uint64)), _ptr(pile.fwd_depth), _ptr(pile.rev_depth),
                            _ptr(pile.fwd_nk), _ptr(pile.rev_nk), _ptr(pile.stats), _ptr(pile.present),
                            _ptr(pile.kmc_stats))
    return pile


def sample_pileup_mt(ix, mates, n_threads, n_fixed=2, use_full_kmer=False, ci=3, cs=1000000, cx=1000000000):
    """Same result as sample_pileup on n_threads host threads (bronko_oracle_mt.c).  mates: list of either lists of ASCII reads
    or 2-D uint8 arrays [n][len] of ASCII symbols.  Returns (Pileup, (stage-1 seconds, stage-2 seconds))."""
    pile = Pileup(ix, len(mates))
    mp = MapParams(n_fixed, int(use_full_kmer), ci, cs, cx)
    flats, lens = [], []
    for m in mates:
        if isinstance(m, np.ndarray) and m.ndim == 2:
            flats.append(np.ascontiguousarray(m, np.uint8).reshape(-1))
            lens.append(np.full(m.shape[0], m.shape[1], np.uint64))
        else:
            flats.append(np.frombuffer(b"".join(bytes(r) for r in m), np.uint8))
            lens.append(np.array([len(r) for r in m], np.uint64))
    flat = np.concatenate(flats + [np.zeros(1, np.uint8)])
    offs = np.zeros(sum(len(x) for x in lens) + 1, np.uint64)
    offs[1:] = np.cumsum(np.concatenate(lens)) if len(offs) > 1 else 0
    moff = np.zeros(len(mates) + 1, np.uint64)
    moff[1:] = np.cumsum([len(x) for x in lens])
    secs = (C.c_double * 2)()
    lib().orc_sample_pileup_mt(ix.h, C.byref(mp), len(mates), flat.ctypes.data, offs.ctypes.data, moff.ctypes.data, int(n_threads),
                               _ptr(pile.fwd_depth), _ptr(pile.rev_depth), _ptr(pile.fwd_nk), _ptr(pile.rev_nk),
                               _ptr(pile.stats), _ptr(pile.present), _ptr(pile.kmc_stats), secs)
    return pile, (secs[0], secs[1])


def pick_best_genome(ix, stats, present):
    stats = np.ascontiguousarray(stats, np.uint64)
    present = np.ascontiguousarray(present, np.uint8)
    return lib().orc_pick_best_genome(ix.h, _ptr(stats), _ptr(present))


def default_call_params(k=21):
    p = CallParams()
    lib().orc_call_params_default(C.byref(p))
    p.k = k
    return p


def baseline_noise(fwd4, rev4):
    fwd4 = np.ascontiguousarray(fwd4, np.uint64)
    rev4 = np.ascontiguousarray(rev4, np.uint64)
    n = len(fwd4) // 4
    mx, mean, sd = np.zeros(n), np.zeros(n), np.zeros(n)
    lib().orc_baseline_noise(_ptr(fwd4), _ptr(rev4), n, _ptr(mx), _ptr(mean), _ptr(sd))
    return mx, mean, sd


def call_variants(ix, file_id, pile, params):
    out = C.POINTER(VcfRecord)()
    nmaj, nmin = C.c_uint64(), C.c_uint64()
    br, dc = C.c_double(), C.c_double()
    n = lib().orc_call_variants(ix.h, file_id, C.byref(params), _ptr(pile.fwd_depth), _ptr(pile.rev_depth),
                                _ptr(pile.fwd_nk), _ptr(pile.rev_nk), C.byref(out), C.byref(nmaj), C.byref(nmin),
                                C.byref(br), C.byref(dc))
    recs = [dict((f[0], getattr(out[i], f[0])) for f in VcfRecord._fields_) for i in range(n)]
    return recs, out, n, nmaj.value, nmin.value, br.value, dc.value


def write_vcf(path, reads_path, ix, file_id, recs_ptr, n):
    if lib().orc_write_vcf(path.encode(), reads_path.encode(), ix.h, file_id, recs_ptr, n) != 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())


def write_pileup(path, ix, file_id, pile):
    if lib().orc_write_pileup(path.encode(), ix.h, file_id, _ptr(pile.fwd_depth), _ptr(pile.rev_depth)) != 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
