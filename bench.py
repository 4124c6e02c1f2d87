#!/usr/bin/env python3
"""bench.py -- reads/sec through the k-mer -> pileup hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic reads that is already resident in HBM:
    bk_sample_begin (zero counters + pileups)  ->  bk_push_reads_packed_device (scan_count + level2 + fold kernels)
    [-> RCCL reduce-scatter(sum) of the k-mer counter plane when world_size > 1]  ->  bk_sample_finalize (thresholds
    + map_kmers kernels).  Outputs stay in HBM.
Steps are independent samples (call.rs:212 handles a run's samples one after the other): `--in-flight` engines on the same
device tables (bk_engine_fork) take them in turn, each on its own stream, so that a sample's scan runs next to the previous
samples' finalize kernels.  value = reads of all K steps / wall time of the K steps; serial_ms_per_step reports the same
steps with one sample at a time.
Workload at N=1: BASELINE.json configs[1] -- SARS-CoV-2 single reference (wuhan_ref.fasta, k=21), 1,000,000
synthetic 150 bp single-end reads (seed 2).  N>1: every rank scans its own 1M-read shard of one sample (weak
scaling), the counter plane is reduce-scattered once per step, every rank maps its part.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--in-flight E] [--cpu-sample S | --no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
ALGO_BYTES_PER_READ = 40    # 150 bases x 2 bit, padded to a 40 B record, read once


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=1000000, help="reads per pass timed through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-passes", type=int, default=4, help="passes of the CPU oracle over the sample (about 10 s in total)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allreduce", action="store_true", help="N > 1: all-reduce the counter plane instead of reduce-scatter + sharded finalize")
    ap.add_argument("--ref-len", type=int, default=0, help="experiment: truncate the reference to its first N bases")
    ap.add_argument("--wide", action="store_true", help="N > 1: move the counter plane as 64-bit integers even when 32 bits would do")
    ap.add_argument("--in-flight", type=int, default=3, help="samples in flight per GPU: steps alternate over this many engines "
                    "(bk_engine_fork: shared index tables, own counter planes / outputs / stream), so that a sample's scan overlaps "
                    "the previous sample's finalize; 1 = strictly one sample after the other")
    ap.add_argument("--backend", default="nccl", help="testing aid: 'gloo' lets several ranks share one GPU (rank r uses GPU r mod #GPUs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from bronko_amd import Params, synth
    from bronko_amd.dist import ShardedFinalize, allreduce_counters
    from bronko_amd.hostlib import HostIndex

    k = 21
    ref_path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
    ref = synth.read_fasta_bytes(ref_path)
    if args.ref_len:
        ref = ref[:args.ref_len]
        ix = HostIndex.build_mem(k, [("wuhan_ref", [("trunc", ref)])], threads=4)
    else:
        ix = HostIndex.build(k, [ref_path], threads=4)
    eng = ix.engine(Params(device=local_rank))

    # synthetic sample: reference + 20 SNPs + 20 iSNVs, 0.5 % substitution errors
    genome, isnv = synth.sample_genome(ref, 2)
    codes = synth.single_end_codes(genome, args.reads, args.read_len, 2 * 1000003 + rank, err=0.005, isnv=isnv)
    words, lens = synth.pack_codes(codes)
    stride = words.shape[1]
    d_words = torch.from_numpy(words.view(np.int32)).to(dev)
    d_lens = torch.from_numpy(lens.view(np.int16)).to(dev)
    n_rec = len(lens)

    # Samples are independent (call.rs:212 handles them one after the other), so `--in-flight` engines take the steps in turn.
    # Each engine launches on its own HIP stream (created with the engine); torch sees it as an ExternalStream and every
    # torch / torch.distributed operation on the engine's buffers is enqueued on it: the collectives are ordered against the
    # kernels by the stream.  (torch's pool streams are not used: on this ROCm two of them may share a hardware queue, and then
    # nothing overlaps.)
    n_fly = max(1, args.in_flight)
    engs = [eng] + [eng.fork() for _ in range(n_fly - 1)]
    streams = [torch.cuda.ExternalStream(e.stream_ptr(), device=dev) for e in engs]
    torch.cuda.set_stream(streams[0])
    counters = [torch.as_tensor(_DevArray(e.counters_ptr(0), e.counter_len, "<i8"), device=dev) for e in engs]

    # N > 1: reduce-scatter of the counter plane + each rank maps its part + max / sum of the small pileups (the cheap form,
    # include/bronko_hip.h); --allreduce selects the plain form (all-reduce the plane, every rank maps everything)
    sharded = world > 1 and not args.allreduce and 64 % world == 0
    # (the plane travels as 32-bit integers when no k-mer of the whole sample can occur 2^31 times)
    narrow = world * args.reads * max(args.read_len - k + 1, 1) < 2 ** 31 and not args.wide
    shard_fin = [ShardedFinalize(e, 1, rank, world, dev, narrow=narrow) for e in engs] if sharded else None

    def step(i):
        j = i % len(engs)
        e = engs[j]
        with torch.cuda.stream(streams[j]):
            e.sample_begin()
            e.push_reads_device(0, d_words.data_ptr(), stride, d_lens.data_ptr(), n_rec)
            if sharded:
                shard_fin[j]()   # RCCL over xGMI: reduce-scatter(sum) + 3 small all-reduces
                return
            if world > 1:
                e.counters_ptr(0)                 # (same pointer every step; a plane nothing was pushed to is zeroed by this call)
                allreduce_counters(counters[j])   # RCCL over xGMI: ONE all-reduce(sum) of the u64 k-mer occurrence counters
            e.sample_finalize(1)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timing(kind):
        for e in engs:
            e.timing_enable(kind)

    def timing_read():
        tot_ms, tot_n = None, None
        for e in engs:
            ms, n = e.timing_read(reset=True)
            tot_ms = list(ms) if tot_ms is None else [a + b for a, b in zip(tot_ms, ms)]
            tot_n = list(n) if tot_n is None else [a + b for a, b in zip(tot_n, n)]
        return tot_ms, tot_n

    for i in range(args.warmup):
        step(i)
    fence()
    # timed region: only the dominant kernel is bracketed by HIP events (on its launch stream) -- bracketing every launch
    # costs ~10 event records per sample; the other kernels' averages come from a short extra pass after the timed region
    timing(2)
    timing_read()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    kms, kn = timing_read()
    timing(1)
    for i in range(max(2, min(6, args.steps))):
        step(i)
    fence()
    kms_all, kn_all = timing_read()
    timing(0)
    kms = [kms[0]] + list(kms_all[1:])
    kn = [kn[0]] + list(kn_all[1:])
    # The same steps strictly one after the other on one engine: a single sample's turnaround, and the dominant kernel's own
    # duration (HIP events on its launch stream) with nothing running next to it -- the figure the roofline object is about;
    # in the timed region above a scan shares the CUs with the other samples' finalize kernels.
    n_serial = max(2, min(10, args.steps))
    saved = engs
    engs = engs[:1]
    step(0)
    fence()
    timing(2)
    timing_read()
    ts0 = time.perf_counter()
    for i in range(n_serial):
        step(i)
    fence()
    serial_ms = (time.perf_counter() - ts0) / n_serial * 1e3
    kms_solo, kn_solo = timing_read()
    timing(0)
    engs = saved
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    res = engs[(args.steps - 1) % len(engs)].sample_download(1, arrays=False)   # sanity only: the last timed step really produced a pileup
    total_reads = args.reads * world * args.steps
    value = total_reads / dt

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure is the
    # one measured with rocprofv3 (separate --pmc passes) on this workload and committed under profiles/
    traffic = None
    valu_insts = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tj.get("workload_reads") == args.reads and tj.get("read_len") == args.read_len and not args.ref_len:
            traffic = tj["traffic_bytes_per_launch"]
            valu_insts = tj.get("valu_wave_insts_per_launch")
    except (OSError, ValueError, KeyError):
        pass

    scan_ms_fly = kms[0] / max(kn[0], 1)
    scan_ms = kms_solo[0] / max(kn_solo[0], 1)
    fin_ms = kms[1] / max(kn[1], 1)
    achieved = (ALGO_BYTES_PER_READ * args.reads) / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    out = {
        "metric": "reads/sec through call k-mer->pileup, SARS-CoV-2 k=21",
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "serial_ms_per_step": serial_ms,   # one sample at a time on one engine (not the headline: see config.samples_in_flight)
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: SARS-CoV-2 single ref (wuhan_ref, 29903 bp), k=21, n_fixed=2, "
                               "%d synthetic %d bp single-end reads per GPU per step, 0.5%% substitution errors, seed 2"
                               % (args.reads, args.read_len),
                   "reads_per_gpu": args.reads, "read_len": args.read_len, "k": k, "samples_in_flight": len(engs),
                   "parallelism": ("reads sharded over %d GPU(s); RCCL reduce-scatter(sum) of the k-mer counter plane" + (" as int32" if narrow else "") + ", sharded finalize, "
                                   "all-reduce(max / sum) of the pileups" if sharded else
                                   "reads sharded over %d GPU(s); RCCL all-reduce(sum) of k-mer counters") % world
                   if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "kernel": "scan_count_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "avg_kernel_ms": scan_ms, "launches": kn_solo[0],
                     "measured": "HIP events around the kernel, steps run one sample at a time after the timed region",
                     "avg_kernel_ms_in_flight": scan_ms_fly, "launches_in_flight": kn[0],   # timed region: sharing the CUs with other samples' kernels
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_READ * args.reads,
                     # the kernel is VALU-issue bound, not HBM bound: wave-instructions per launch from
                     # the committed PMC pass; a wave64 VALU instruction occupies one of the chip's 1024 SIMDs for 4 cycles
                     "valu_wave_insts_per_launch": valu_insts,
                     "valu_busy_frac": (valu_insts * 4 / (1024 * 2.4e9) / (scan_ms * 1e-3)) if valu_insts and scan_ms > 0 else None},
        "kernels_ms": {"scan_count": scan_ms, "finalize": fin_ms, "memset_copy": kms[2] / max(kn[2], 1),
                       "level2_fold": kms[3] / max(kn[3], 1)},   # (averages of the in-flight pass: kernels share the chip)
        "check": {"perfect_kmers": int(res.stats[0, 0, 0]), "variant_kmers": int(res.stats[0, 0, 1]),
                  "kmers_scanned": int(res.kmer_stats[0, 1])},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline = the oracle (literal single-threaded C restatement of bronko v0.1.0: exact k-mer counting
        # standing in for KMC3, then map_kmers) on a bounded sample of the same workload.  Checker code, timed
        # here only as the reported baseline -- it is never on the product path.
        from oracle import oracle as orc
        n_s = min(args.cpu_sample, args.reads)
        sample = synth.codes_to_ascii(codes[:n_s])
        oix = orc.Index.build(k, [ref_path])
        c0 = time.perf_counter()
        for _ in range(args.cpu_passes):
            orc.sample_pileup(oix, [sample])
        cdt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": n_s * args.cpu_passes / cdt, "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": "%d passes over the first %d reads of the same batch: oracle exact k-mer counting "
                                         "(KMC3 stand-in) + map_kmers, single thread, %.1f s in total; host has %d cores"
                                         % (args.cpu_passes, n_s, cdt, os.cpu_count() or 0)}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    for e in reversed(engs):
        e.close()


if __name__ == "__main__":
    main()
