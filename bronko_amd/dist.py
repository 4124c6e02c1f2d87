"""Multi-GPU plumbing for one sample sharded over ranks.

One process per GPU; the only exchange step is ONE collective over the k-mer occurrence counter plane per (sample, mate file),
between the last push and finalize: an all-reduce(sum) of the u64 plane, or -- the cheap form, ShardedFinalize -- a
reduce-scatter(sum) of the plane packed by the engine to 16- / 32-bit elements followed by a sharded finalize.  The thresholds (-ci/-cs/-cx) and the max / distinct-count
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


def _all_reduce(t, op, group=None):
    """dist.all_reduce in place; with the gloo backend (CPU tests, or several test ranks sharing one GPU) device tensors
    are staged through host memory explicitly.  group: the process group to use (None: the default one) -- a host with several
    samples in flight gives every engine a group of its own (its own RCCL communicator and stream), so that the collectives of
    one sample do not queue behind another's."""
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _active(force=False):
    """Is there a process group with someone to talk to?  force: yes even with one rank -- the collectives then run through the
    backend all the same (tests/test_gpu_dist.py: first contact with RCCL on a one-GPU box)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)


def allreduce_counters(counters, force=False, group=None):
    """In-place sum of a counter plane across ranks.  `counters`: int64 view of the engine's u64 plane
    (two's-complement addition is the same operation); a no-op outside a process group."""
    if _active(force):
        assert counters.dtype == torch.int64
        _all_reduce(counters, dist.ReduceOp.SUM, group)
    return counters


def combine_shard_results(depth, nk, sums, force=False, group=None):
    """After every rank mapped its part of the planes: depth = max over ranks, #k-mers and the small statistics add."""
    if _active(force):
        _all_reduce(depth, dist.ReduceOp.MAX, group)
        _all_reduce(nk, dist.ReduceOp.SUM, group)
        _all_reduce(sums, dist.ReduceOp.SUM, group)


def _all_to_all(out, inp, out_splits, in_splits, group=None):
    """dist.all_to_all_single; staged through host memory with the gloo backend (CPU tests / several test ranks on one GPU)."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(ho, hi, out_splits, in_splits, group=group)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def exchange_entries(keys, cnts, in_splits, group=None):
    """All-to-all of (key, count) entries grouped by destination rank: `in_splits[r]` consecutive entries of `keys` (int64) /
    `cnts` (int32) go to rank r.  Returns what this rank received (keys, counts), groups in rank order."""
    world = dist.get_world_size(group)
    t_in = torch.tensor(in_splits, dtype=torch.int64, device=keys.device)
    t_out = torch.empty(world, dtype=torch.int64, device=keys.device)
    _all_to_all(t_out, t_in, [1] * world, [1] * world, group)
    out_splits = [int(x) for x in t_out.tolist()]
    n_recv = sum(out_splits)
    rk = torch.empty(n_recv, dtype=torch.int64, device=keys.device)
    rc = torch.empty(n_recv, dtype=torch.int32, device=keys.device)
    _all_to_all(rk, keys, out_splits, in_splits, group)
    _all_to_all(rc, cnts, out_splits, in_splits, group)
    return rk, rc


def exchange_kmer_tables(eng, rank, world, device, force=False, group=None):
    """full_kmer_stats with one sample's reads sharded over ranks (include/bronko_hip.h, bk_kmer_table_partition): every k-mer
    that touches no window bucket is moved to its owner rank (a hash of the key) -- one all-to-all of (key u64, count u32)
    entries, the one real exchange step of KMC's distinct / counted totals -- where equal keys add up.  Called between the last
    push and the sharded finalize (ShardedFinalize does it itself when the engine has the statistics table).  Returns the
    received tensors: the caller keeps them until the engine's stream has consumed them."""
    if not _active(force):
        return None
    kp, cp, off = eng.kmer_table_partition(world)
    n_send = off[world]
    keys = torch.as_tensor(DeviceVector(kp, max(n_send, 1)), device=device)[:n_send]
    cnts = torch.as_tensor(DeviceVector(cp, max(n_send, 1), "<i4"), device=device)[:n_send]
    rk, rc = exchange_entries(keys, cnts, [off[r + 1] - off[r] for r in range(world)], group)
    eng.kmer_table_replace(rk.data_ptr() if len(rk) else 0, rc.data_ptr() if len(rc) else 0, len(rk))
    return rk, rc


class DeviceVector:
    """__cuda_array_interface__ view of n little-endian integers (64-bit unless typestr says otherwise) at a device pointer (for
    torch.as_tensor)."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def pick_width(max_e, max_v, world):
    """Narrowest transport width (bits) at which a reduce-scatter over `world` ranks is exact, given the largest E count and the
    largest |V element| over all ranks' planes (include/bronko_hip.h, bk_shard_measure)."""
    if max_v * world <= 32767 and max_e < 2 ** 32:
        return 16
    if max(max_e, max_v) * world <= 2 ** 31 - 1:
        return 32
    return 64


# element type on the wire: width 16 travels as int32 words of two 16-bit lanes each (RCCL has no 16-bit integer type; the lanes
# are unsigned and their sums stay below 2^16, so they add up inside 32-bit additions -- bk_kernels.hip, xport_pack_kernel)
_WIDTH_DTYPE = {16: (torch.int32, "<i4", 4), 32: (torch.int32, "<i4", 4), 64: (torch.int64, "<i8", 8)}


def reduce_scatter_typed(send, recv, rank, world, force=False, group=None):
    """Sum `send` (world equal parts) across ranks, leaving part `rank` in `recv`.  RCCL reduce-scatter on GPUs; gloo (CPU tests,
    several test ranks on one GPU) has none: all-reduce of a copy, then the part.  force: issue the collective even with one rank
    (the first-contact test of the RCCL branch on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (world == 1 and not force):
        recv.copy_(send[rank * recv.numel():(rank + 1) * recv.numel()])
        return
    if dist.get_backend(group) == "gloo":
        tmp = send.cpu() if send.is_cuda else send.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        recv.copy_(tmp[rank * recv.numel():(rank + 1) * recv.numel()])
        return
    dist.reduce_scatter_tensor(recv, send, op=dist.ReduceOp.SUM, group=group)


class ShardedFinalize:
    """The cheap multi-GPU form (include/bronko_hip.h): the engine packs each counter plane for the wire (bk_shard_transport: 16-,
    32- or 64-bit elements), ONE reduce-scatter(sum) per mate file moves it, the engine widens the received part
    (bk_shard_received), this rank maps it (bk_sample_finalize_shard), the small pileups are combined (max / sum) and the
    statistics summed.  After __call__ eng.sample_download() returns the full result on every rank.
    width: 16 / 32 / 64, or "auto" = measure the planes first (bk_shard_measure + a two-word all-reduce(max) + one host
    synchronisation per mate file) and take the narrowest exact width.  A fixed width that turns out too narrow is detected on
    the device and reported by sample_download / eng.transport_overflow() on every rank -- never silently wrong.
    The engine must run on torch's current stream (torch.cuda.stream(ExternalStream(eng.stream_ptr()))): the collectives are
    ordered against the engine's kernels by that stream only."""

    def __init__(self, eng, n_mates, rank, world, device, width="auto", force_collectives=False, time_comm=False, group=None, local_only=False):
        self.eng, self.n_mates, self.rank, self.world, self.width, self.device = eng, n_mates, rank, world, width, device
        self.force = force_collectives
        self.group = group            # process group of this engine's collectives (None: the default group)
        self.local_only = local_only  # measurement aid: every collective replaced by its local stand-in (wrong results, same kernels)
        cells4 = eng.total_cells * 4
        pile = torch.as_tensor(DeviceVector(eng.pileup_ptr(), 4 * cells4), device=device)
        self.depth, self.nk = pile[:2 * cells4], pile[2 * cells4:]
        sp, sn = eng.shard_sums()
        self.sums = torch.as_tensor(DeviceVector(sp, sn), device=device)
        self._views = {}          # (pointer, elements, width) -> tensor view of an engine buffer
        self.last_widths = []     # widths the last sample's mate files travelled at
        self.time_comm = time_comm
        self._events = []         # (start, end) around the collectives of each sample
        self.bytes_sent = 0       # transport bytes of the planes handed to the reduce-scatter (all parts), last sample

    def _view(self, ptr, n, width):
        key = (ptr, n, width)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = torch.as_tensor(DeviceVector(ptr, n, _WIDTH_DTYPE[width][1]), device=self.device)
        return v

    def _active(self):
        return _active(self.force) and not self.local_only

    def _measure(self, m):
        t = self._view(self.eng.shard_measure(m), 2, 64)
        if self._active():
            _all_reduce(t, dist.ReduceOp.MAX, self.group)
        mx = t.tolist()          # (synchronises: the host needs the maxima to choose)
        return pick_width(int(mx[0]), int(mx[1]), self.world)

    def __call__(self):
        if self.depth.is_cuda and torch.cuda.current_stream().cuda_stream != self.eng.stream_ptr():
            raise RuntimeError("ShardedFinalize: torch's current stream is not the engine's stream -- the collectives would not be "
                               "ordered against the engine's kernels (use torch.cuda.stream(ExternalStream(eng.stream_ptr())))")
        if self.eng.full_kmer_stats and self._active():
            self._held = exchange_kmer_tables(self.eng, self.rank, self.world, self.device, self.force, self.group)   # KMC's distinct / counted totals stay exact
        ev = None
        if self.time_comm:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self.last_widths, self.bytes_sent = [], 0
        for m in range(self.n_mates):
            w = self._measure(m) if self.width == "auto" else self.width
            try:
                send_ptr, part_bytes, recv_ptr = self.eng.shard_transport(m, self.world, w)
            except RuntimeError:   # (engine.BronkoError)
                # (the engine refuses width 16 where it would not shrink the plane -- many shards, every E count four lanes --; every
                # rank sees the same geometry, so every rank falls back the same way)
                if self.width != "auto" or w != 16:
                    raise
                w = 32
                send_ptr, part_bytes, recv_ptr = self.eng.shard_transport(m, self.world, w)
            self.last_widths.append(w)
            item = _WIDTH_DTYPE[w][2]
            send = self._view(send_ptr, part_bytes // item * self.world, w)
            recv = self._view(recv_ptr, part_bytes // item, w)
            self.bytes_sent += part_bytes * self.world
            if ev and m == 0:
                ev[0].record()
            if self.local_only:
                recv.copy_(send[self.rank * recv.numel():(self.rank + 1) * recv.numel()])
            else:
                reduce_scatter_typed(send, recv, self.rank, self.world, self.force, self.group)
            if ev and m == self.n_mates - 1:
                ev[1].record()
            self.eng.shard_received(m, self.rank, self.world, w)
        self.eng.sample_finalize_shard(self.n_mates, self.rank, self.world)
        if ev:
            ev[2].record()
        if not self.local_only:
            combine_shard_results(self.depth, self.nk, self.sums, self.force, self.group)
        if ev:
            ev[3].record()
            self._events.append(ev)
        self.eng.sample_merge_shards()

    def comm_ms(self, reset=True):
        """(reduce-scatter ms, combine ms, samples) accumulated by time_comm since the last reset; synchronises."""
        torch.cuda.synchronize()
        rs = sum(e[0].elapsed_time(e[1]) for e in self._events)
        cb = sum(e[2].elapsed_time(e[3]) for e in self._events)
        n = len(self._events)
        if reset:
            self._events = []
        return rs, cb, n


def sharded_finalize(eng, n_mates, rank, world, device, width="auto"):
    """One-shot convenience wrapper of ShardedFinalize."""
    ShardedFinalize(eng, n_mates, rank, world, device, width=width)()
