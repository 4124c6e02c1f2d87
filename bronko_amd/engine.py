"""Python mirror of the C ABI (include/bronko_hip.h): same call order, same argument meaning, same errors.

    eng = Engine(k, bucket_ids, bucket_off, entries, files, Params(...))   # = bk_engine_create
    eng.sample_begin()                                                      # = initialize_output_maps
    eng.push_reads(mate, words, lens)                                       # = reads of one mate file
    res = eng.sample_finish(n_mates)                                        # = KMC thresholds + map_kmers

A non-zero status raises BronkoError with bk_last_error(); the reference's convention for the same
conditions is `error!(..); exit(1)`.
"""
import ctypes as C

import numpy as np

from . import _ffi

BUCKET_INFO_DTYPE = np.dtype({"names": ["file_id", "seq_id", "location", "idx", "canonical"],
                              "formats": [np.uint16, np.uint8, np.uint32, np.uint8, np.uint8],
                              "offsets": [0, 2, 4, 8, 9], "itemsize": 12})


class BronkoError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("bronko_hip status %d: %s" % (status, msg))
        self.status = status


def lib_path():
    return _ffi.LIB_PATH


def _check(rc, L=None):
    if rc != 0:
        raise BronkoError(rc, (L or _ffi.load()).bk_last_error().decode(errors="replace"))


def Params(n_fixed=2, use_full_kmer=False, ci=3, cs=1000000, cx=1000000000, device=0, full_kmer_stats=False,
           kmer_table_log2=None, pileup_selected_only=False):
    p = _ffi.Params()
    _ffi.load().bk_params_default(C.byref(p))
    p.n_fixed, p.use_full_kmer, p.ci, p.cs, p.cx, p.device = n_fixed, int(use_full_kmer), ci, cs, cx, device
    p.full_kmer_stats = int(full_kmer_stats)
    p.pileup_selected_only = int(pileup_selected_only)
    if kmer_table_log2 is not None:
        p.kmer_table_log2 = kmer_table_log2
    return p


def pack_reads(reads, k, stride_words=None):
    """K0 (bk_pack_reads): list of ASCII reads -> (words u32[n][stride], lens u16[n]) fixed-stride 2-bit records."""
    L = _ffi.load()
    reads = [bytes(r) for r in reads]
    if stride_words is None:
        longest = max([len(r) for r in reads] + [k])
        stride_words = min((longest + 15) // 16, 4095)
    flat = np.frombuffer(b"".join(reads), np.uint8) if reads else np.zeros(0, np.uint8)
    flat = np.ascontiguousarray(flat) if len(flat) else np.zeros(1, np.uint8)
    off = np.zeros(len(reads) + 1, np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads])
    n = L.bk_pack_reads_flat(flat.ctypes.data, off.ctypes.data, len(reads), k, stride_words, None, None, 0)
    words = np.zeros((max(n, 1), stride_words), np.uint32)
    lens = np.zeros(max(n, 1), np.uint16)
    L.bk_pack_reads_flat(flat.ctypes.data, off.ctypes.data, len(reads), k, stride_words, words.ctypes.data,
                         lens.ctypes.data, n)
    return words[:n], lens[:n]


def build_index_device(k, files, device=0):
    """bk_build_index: build_indexes (build.rs:145-231) on the GPU.  files = [(file_name, [(seq_name, seq_bytes), ...]), ...];
    returns (bucket_ids u64[n], bucket_off u64[n + 1], entries BucketInfo[m]) -- what Engine() takes."""
    L = _ffi.load()
    n_seqs = np.array([len(f[1]) for f in files] + [0], np.int32)
    seqs = [bytes(s[1]) for f in files for s in f[1]]
    seq_lens = np.array([len(s) for s in seqs] + [0], np.uint64)
    bufs = [C.create_string_buffer(s, max(len(s), 1)) for s in seqs]
    ptrs = (C.c_void_p * max(len(bufs), 1))(*[C.addressof(b) for b in bufs])
    out = _ffi.BuiltIndex()
    rc = L.bk_build_index(k, len(files), n_seqs.ctypes.data, seq_lens.ctypes.data, C.addressof(ptrs), device, C.byref(out))
    if rc != 0:
        raise BronkoError(rc, L.bk_build_last_error().decode(errors="replace"))
    try:
        nb, ne = out.n_buckets, out.n_entries
        ids = np.ctypeslib.as_array(C.cast(out.bucket_ids, C.POINTER(C.c_uint64)), shape=(nb,)).copy() if nb else np.zeros(0, np.uint64)
        off = np.ctypeslib.as_array(C.cast(out.bucket_off, C.POINTER(C.c_uint64)), shape=(nb + 1,)).copy()
        ent = (np.ctypeslib.as_array(C.cast(out.entries, C.POINTER(C.c_uint8)), shape=(ne * 12,)).copy().view(BUCKET_INFO_DTYPE)
               if ne else np.zeros(0, BUCKET_INFO_DTYPE))
    finally:
        L.bk_built_index_free(C.byref(out))
    return ids, off, ent


class SampleResult:
    """Outputs of one sample: the four OutputData arrays (call.rs:1235-1239,1451-1454) + map_kmers' stats."""

    def __init__(self, n_mates, n_files, total_cells):
        n = total_cells * 4
        self.fwd_depth = np.zeros(n, np.uint64)
        self.rev_depth = np.zeros(n, np.uint64)
        self.fwd_nk = np.zeros(n, np.uint64)
        self.rev_nk = np.zeros(n, np.uint64)
        self.stats = np.zeros((n_mates, n_files, 3), np.uint64)
        self.present = np.zeros((n_mates, n_files), np.uint8)
        self.kmer_stats = np.zeros((n_mates, 4), np.uint64)

    def arrays(self):
        return self.fwd_depth, self.rev_depth, self.fwd_nk, self.rev_nk


def device_memory(device=0):
    """(free, total) bytes of a device (bk_device_memory): what the CLI sizes its number of ingest lanes by."""
    L = _ffi.load()
    f, t = C.c_uint64(), C.c_uint64()
    _check(L.bk_device_memory(device, C.byref(f), C.byref(t)), L)
    return f.value, t.value


class Engine:
    def __init__(self, k, bucket_ids, bucket_off, entries, files, params=None):
        """files: [(file_name, [(seq_name, seq_bytes), ...]), ...] = ViralMetadata (build.rs:46-50)"""
        L = _ffi.load()
        self._L = L
        self.h = None
        params = params or Params()
        ids = np.ascontiguousarray(bucket_ids, np.uint64)
        off = np.ascontiguousarray(bucket_off, np.uint64)
        ent = np.ascontiguousarray(entries)
        if ent.dtype != BUCKET_INFO_DTYPE:
            raise TypeError("entries must have the 12-byte BucketInfo dtype")
        n_seqs = np.array([len(f[1]) for f in files] + [0], np.int32)
        seqs = [bytes(s[1]) for f in files for s in f[1]]
        seq_lens = np.array([len(s) for s in seqs] + [0], np.uint64)
        bufs = [C.create_string_buffer(s, max(len(s), 1)) for s in seqs]
        ptrs = (C.c_void_p * max(len(bufs), 1))(*[C.addressof(b) for b in bufs])
        d = _ffi.IndexDesc(k, len(ids), ids.ctypes.data, off.ctypes.data, ent.ctypes.data, len(ent), len(files),
                           n_seqs.ctypes.data, seq_lens.ctypes.data, C.addressof(ptrs))
        h = C.c_void_p()
        _check(L.bk_engine_create(C.byref(d), C.byref(params), C.byref(h)), L)
        self.h = h
        self.k = k
        self.params = params
        self.n_files = L.bk_n_files(h)
        self.total_cells = L.bk_total_cells(h)
        self.n_slots = L.bk_n_slots(h)
        self.counter_len = L.bk_counter_len(h)

    def fork(self, params=None):
        """A second engine on the same device tables with its own counter planes, outputs and stream (bk_engine_fork):
        alternate independent samples over the two so that one's scan overlaps the other's finalize.  params: the fork's own
        ci / cs / cx / pileup_selected_only (bk_engine_fork_params; what shapes the tables must equal the parent's)."""
        h = C.c_void_p()
        if params is None:
            _check(self._L.bk_engine_fork(self.h, C.byref(h)), self._L)
        else:
            _check(self._L.bk_engine_fork_params(self.h, C.byref(params), C.byref(h)), self._L)
        e = object.__new__(Engine)
        e._L, e.h, e.k, e.params = self._L, h, self.k, params if params is not None else self.params
        e.n_files, e.total_cells, e.n_slots, e.counter_len = self.n_files, self.total_cells, self.n_slots, self.counter_len
        e._parent = self   # the parent's tables must outlive the fork
        return e

    def close(self):
        if self.h:
            self._L.bk_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(self._L.bk_engine_set_stream(self.h, C.c_void_p(stream_ptr)), self._L)

    def stream_ptr(self):
        """hipStream_t (as int) the engine launches on -- wrap it (torch.cuda.ExternalStream) to order other work with it."""
        return int(self._L.bk_engine_get_stream(self.h) or 0)

    def sample_begin(self):
        _check(self._L.bk_sample_begin(self.h), self._L)

    def push_reads(self, mate, words, lens):
        words = np.ascontiguousarray(words, np.uint32)
        lens = np.ascontiguousarray(lens, np.uint16)
        if len(lens) == 0:
            return
        assert words.ndim == 2 and words.shape[0] == len(lens)
        _check(self._L.bk_push_reads_packed(self.h, mate, words.ctypes.data, words.shape[1], lens.ctypes.data, len(lens)), self._L)

    def push_reads_ascii(self, mate, reads):
        """bk_push_reads_ascii: list of ASCII reads, packed on the GPU, asynchronous."""
        reads = [bytes(r) for r in reads]
        if not reads:
            return
        flat = np.frombuffer(b"".join(reads), np.uint8)
        flat = np.ascontiguousarray(flat) if len(flat) else np.zeros(1, np.uint8)
        off = np.zeros(len(reads) + 1, np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads])
        _check(self._L.bk_push_reads_ascii(self.h, mate, flat.ctypes.data, off.ctypes.data, len(reads)), self._L)

    def push_reads_ascii_device(self, mate, d_bases_ptr, d_offsets_ptr, n_reads, total_bases, longest_read):
        """bk_push_reads_ascii_device: sequence lines resident in device memory, packed (K0) and scanned on the engine's stream."""
        _check(self._L.bk_push_reads_ascii_device(self.h, mate, C.c_void_p(d_bases_ptr), C.c_void_p(d_offsets_ptr), n_reads, total_bases,
                                                  longest_read), self._L)

    def push_reads_device(self, mate, d_words_ptr, stride_words, d_lens_ptr, n_records):
        _check(self._L.bk_push_reads_packed_device(self.h, mate, C.c_void_p(d_words_ptr), stride_words,
                                                   C.c_void_p(d_lens_ptr), n_records), self._L)

    def counters_ptr(self, mate):
        p = C.c_void_p()
        _check(self._L.bk_counters_device_ptr(self.h, mate, C.byref(p)), self._L)
        return p.value

    def pileup_ptr(self):
        p = C.c_void_p()
        _check(self._L.bk_pileup_device_ptr(self.h, C.byref(p)), self._L)
        return p.value

    def sample_finalize(self, n_mates=1):
        _check(self._L.bk_sample_finalize(self.h, n_mates), self._L)

    def sample_finalize_shard(self, n_mates, shard, n_shards):
        """Map only the shard-th of n_shards equal parts of each counter plane (include/bronko_hip.h)."""
        _check(self._L.bk_sample_finalize_shard(self.h, n_mates, shard, n_shards), self._L)

    def shard_measure(self, mate):
        """Device pointer of two u64: the largest E count and the largest |V element| of this rank's plane (asynchronous; the
        host all-reduces them with MAX and picks the transport width: include/bronko_hip.h)."""
        p = C.c_void_p()
        _check(self._L.bk_shard_measure(self.h, mate, C.byref(p)), self._L)
        return p.value

    def shard_transport(self, mate, n_shards, width):
        """Pack the plane for the reduce-scatter: (send pointer, bytes per part, receive pointer); width 16 / 32 / 64 bits."""
        s, r, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._L.bk_shard_transport(self.h, mate, n_shards, width, C.byref(s), C.byref(n), C.byref(r)), self._L)
        return s.value, n.value, r.value

    def shard_received(self, mate, shard, n_shards, width):
        _check(self._L.bk_shard_received(self.h, mate, shard, n_shards, width), self._L)

    def transport_overflow(self):
        """True when some sample since the last call met a counter too large for the width its plane was exchanged at."""
        f = C.c_int(0)
        _check(self._L.bk_transport_overflow(self.h, C.byref(f)), self._L)
        return bool(f.value)

    def kmer_table_partition(self, n_parts):
        """full_kmer_stats under a sharded finalize: (device pointer of keys u64, of counts u32, offsets[n_parts + 1]) -- the
        statistics table's entries grouped by owner rank.  Synchronises."""
        k, c = C.c_void_p(), C.c_void_p()
        off = (C.c_uint64 * (n_parts + 1))()
        _check(self._L.bk_kmer_table_partition(self.h, n_parts, C.byref(k), C.byref(c), off), self._L)
        return k.value, c.value, list(off)

    def kmer_table_replace(self, keys_ptr, counts_ptr, n):
        """... and the table rebuilt from the entries this rank owns (device pointers; equal keys add up)."""
        _check(self._L.bk_kmer_table_replace(self.h, keys_ptr, counts_ptr, n), self._L)

    @property
    def full_kmer_stats(self):
        return bool(self.params.full_kmer_stats)

    def shard_sums(self):
        """(device pointer, u64 length) of the small additive results of sample_finalize_shard."""
        p, n = C.c_void_p(), C.c_uint64()
        _check(self._L.bk_shard_sums_device_ptr(self.h, C.byref(p), C.byref(n)), self._L)
        return p.value, n.value

    def sample_merge_shards(self):
        _check(self._L.bk_sample_merge_shards(self.h), self._L)

    def sample_download(self, n_mates=1, arrays=True):
        r = SampleResult(n_mates, self.n_files, self.total_cells)
        a = [x.ctypes.data if arrays else None for x in (r.fwd_depth, r.rev_depth, r.fwd_nk, r.rev_nk)]
        _check(self._L.bk_sample_download(self.h, n_mates, a[0], a[1], a[2], a[3], r.stats.ctypes.data,
                                          r.present.ctypes.data, r.kmer_stats.ctypes.data), self._L)
        return r

    def sample_finish(self, n_mates=1):
        self.sample_finalize(n_mates)
        return self.sample_download(n_mates)

    def call_params(self, **kw):
        """bk_call_params with the reference's defaults (cli.rs:92-135), k = the engine's; keyword overrides."""
        p = _ffi.CallParams()
        self._L.bk_call_params_default(C.byref(p))
        p.k = self.k
        for name, v in kw.items():
            setattr(p, name, v)
        return p

    def sample_call(self, n_mates=1, params=None):
        """Reference selection + baseline noise + variant calls on the device for the sample just finalized (asynchronous)."""
        _check(self._L.bk_sample_call(self.h, n_mates, C.byref(params or self.call_params())), self._L)

    def download_calls(self):
        """(summary, records) of sample_call: records sorted by (sequence, position, alternative base)."""
        summ = _ffi.CallSummary()
        _check(self._L.bk_sample_download_calls(self.h, C.byref(summ), None, 0), self._L)   # (the summary first: how many records there are)
        cap = max(1, int(summ.n_records))
        recs = (_ffi.CallRecord * cap)()
        _check(self._L.bk_sample_download_calls(self.h, C.byref(summ), recs, cap), self._L)
        return summ, [recs[i] for i in range(min(summ.n_records, cap))]

    def download_noise(self):
        """Noise.max per position of the genome sample_call selected (diagnostic; call.rs:953-962)."""
        n = C.c_uint64(0)
        out = np.zeros(max(1, self.total_cells), np.float64)
        _check(self._L.bk_sample_download_noise(self.h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)), self._L)
        return out[:n.value]

    def timing_enable(self, on=True):
        _check(self._L.bk_timing_enable(self.h, int(on)), self._L)

    def timing_read(self, reset=True):
        ms = (C.c_double * 4)()
        n = (C.c_uint64 * 4)()
        _check(self._L.bk_timing_read(self.h, ms, n, int(reset)), self._L)
        return list(ms), list(n)
