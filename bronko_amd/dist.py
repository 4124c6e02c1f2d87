"""Multi-GPU plumbing for one sample sharded over ranks.

One process per GPU; the only exchange step is ONE all-reduce(sum) of the u64 k-mer occurrence counter plane per
(sample, mate file), between the last push and finalize.  The thresholds (-ci/-cs/-cx) and the max / distinct-count
votes of map_kmers are not linear, so pileups are never reduced -- they are computed after the reduction.
`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI); "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` (reads of one mate file are order-free)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_reduce(t, op):
    """dist.all_reduce in place; with the gloo backend (CPU tests, or several test ranks sharing one GPU) device tensors
    are staged through host memory explicitly."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def allreduce_counters(counters):
    """In-place sum of a counter plane across ranks.  `counters`: int64 view of the engine's u64 plane
    (two's-complement addition is the same operation); a no-op outside a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        assert counters.dtype == torch.int64
        _all_reduce(counters, dist.ReduceOp.SUM)
    return counters


def reduce_scatter_plane(plane, rank, world, out=None, narrow=False):
    """Sum `plane` (1-D int64, length divisible by world) across ranks, leaving only this rank's part -- elements
    [rank * n / world, (rank + 1) * n / world) -- summed, in place; the other parts are left as they were.
    RCCL reduce-scatter on GPUs; gloo (CPU tests) has no reduce-scatter, there it is an all-reduce.
    narrow: move the plane as 32-bit integers (half the bytes over xGMI).  The plane's elements are counts and differences of
    counts that wrap modulo 2^64; truncated to 32 bits, summed modulo 2^32 and sign-extended they are the same numbers as long
    as every true sum lies in [-2^31, 2^31) -- the caller vouches for that (no k-mer of the sample occurs 2^31 times)."""
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return plane
    assert plane.dtype == torch.int64 and plane.numel() % world == 0
    part = plane.numel() // world
    src = plane.to(torch.int32) if narrow else plane
    if dist.get_backend() == "gloo":
        _all_reduce(src, dist.ReduceOp.SUM)
        if narrow:
            plane.copy_(src)   # int32 -> int64 sign-extends
        return plane
    if out is None or out.dtype != src.dtype:
        out = torch.empty(part, dtype=src.dtype, device=plane.device)
    dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM)
    plane[rank * part:(rank + 1) * part].copy_(out)   # (int32 -> int64 sign-extends)
    return plane


def combine_shard_results(depth, nk, sums):
    """After every rank mapped its part of the planes: depth = max over ranks, #k-mers and the small statistics add."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        _all_reduce(depth, dist.ReduceOp.MAX)
        _all_reduce(nk, dist.ReduceOp.SUM)
        _all_reduce(sums, dist.ReduceOp.SUM)


def _all_to_all(out, inp, out_splits, in_splits):
    """dist.all_to_all_single; staged through host memory with the gloo backend (CPU tests / several test ranks on one GPU)."""
    if dist.get_backend() == "gloo" and inp.is_cuda:
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(ho, hi, out_splits, in_splits)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits)


def exchange_entries(keys, cnts, in_splits):
    """All-to-all of (key, count) entries grouped by destination rank: `in_splits[r]` consecutive entries of `keys` (int64) /
    `cnts` (int32) go to rank r.  Returns what this rank received (keys, counts), groups in rank order."""
    world = dist.get_world_size()
    t_in = torch.tensor(in_splits, dtype=torch.int64, device=keys.device)
    t_out = torch.empty(world, dtype=torch.int64, device=keys.device)
    _all_to_all(t_out, t_in, [1] * world, [1] * world)
    out_splits = [int(x) for x in t_out.tolist()]
    n_recv = sum(out_splits)
    rk = torch.empty(n_recv, dtype=torch.int64, device=keys.device)
    rc = torch.empty(n_recv, dtype=torch.int32, device=keys.device)
    _all_to_all(rk, keys, out_splits, in_splits)
    _all_to_all(rc, cnts, out_splits, in_splits)
    return rk, rc


def exchange_kmer_tables(eng, rank, world, device):
    """full_kmer_stats with one sample's reads sharded over ranks (include/bronko_hip.h, bk_kmer_table_partition): every k-mer
    that touches no window bucket is moved to its owner rank (a hash of the key) -- one all-to-all of (key u64, count u32)
    entries, the one real exchange step of KMC's distinct / counted totals -- where equal keys add up.  Called between the last
    push and the sharded finalize (ShardedFinalize does it itself when the engine has the statistics table).  Returns the
    received tensors: the caller keeps them until the engine's stream has consumed them."""
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return None
    kp, cp, off = eng.kmer_table_partition(world)
    n_send = off[world]
    keys = torch.as_tensor(DeviceVector(kp, max(n_send, 1)), device=device)[:n_send]
    cnts = torch.as_tensor(DeviceVector(cp, max(n_send, 1), "<i4"), device=device)[:n_send]
    rk, rc = exchange_entries(keys, cnts, [off[r + 1] - off[r] for r in range(world)])
    eng.kmer_table_replace(rk.data_ptr() if len(rk) else 0, rc.data_ptr() if len(rc) else 0, len(rk))
    return rk, rc


class DeviceVector:
    """__cuda_array_interface__ view of n little-endian integers (64-bit unless typestr says otherwise) at a device pointer (for
    torch.as_tensor)."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class ShardedFinalize:
    """The cheap multi-GPU form (include/bronko_hip.h): reduce-scatter the counter planes, map this rank's part, combine the
    pileups (max / sum) and the statistics.  After __call__ eng.sample_download() returns the full result on every rank.
    The tensor views of the engine's device buffers are made once.
    The engine must run on torch's current stream (eng.set_stream(torch.cuda.current_stream().cuda_stream) with a stream that
    is NOT the default one, whose handle 0 means "the engine's own stream"): the collectives are ordered against the engine's
    kernels by that stream only."""

    def __init__(self, eng, n_mates, rank, world, device, narrow=False):
        self.eng, self.n_mates, self.rank, self.world, self.narrow = eng, n_mates, rank, world, narrow
        self.planes = [torch.as_tensor(DeviceVector(eng.counters_ptr(m), eng.counter_len), device=device) for m in range(n_mates)]
        cells4 = eng.total_cells * 4
        pile = torch.as_tensor(DeviceVector(eng.pileup_ptr(), 4 * cells4), device=device)
        self.depth, self.nk = pile[:2 * cells4], pile[2 * cells4:]
        sp, sn = eng.shard_sums()
        self.sums = torch.as_tensor(DeviceVector(sp, sn), device=device)
        part = eng.counter_len // world
        self.out = torch.empty(part, dtype=torch.int32 if narrow else torch.int64, device=device) if world > 1 else None

    def __call__(self):
        if self.planes and self.planes[0].is_cuda and torch.cuda.current_stream().cuda_stream != self.eng.stream_ptr():
            raise RuntimeError("ShardedFinalize: torch's current stream is not the engine's stream -- the collectives would not be "
                               "ordered against the engine's kernels (use torch.cuda.stream(ExternalStream(eng.stream_ptr())))")
        if self.eng.full_kmer_stats and self.world > 1:
            self._held = exchange_kmer_tables(self.eng, self.rank, self.world, self.depth.device)   # KMC's distinct / counted totals stay exact
        for m, plane in enumerate(self.planes):
            self.eng.counters_ptr(m)   # (a plane this rank pushed nothing to is zeroed by this call)
            reduce_scatter_plane(plane, self.rank, self.world, self.out, self.narrow)
        self.eng.sample_finalize_shard(self.n_mates, self.rank, self.world)
        combine_shard_results(self.depth, self.nk, self.sums)
        self.eng.sample_merge_shards()


def sharded_finalize(eng, n_mates, rank, world, device):
    """One-shot convenience wrapper of ShardedFinalize."""
    ShardedFinalize(eng, n_mates, rank, world, device)()
