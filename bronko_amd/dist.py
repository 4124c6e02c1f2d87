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


def allreduce_counters(counters):
    """In-place sum of a counter plane across ranks.  `counters`: int64 view of the engine's u64 plane
    (two's-complement addition is the same operation); a no-op outside a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        assert counters.dtype == torch.int64
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters
