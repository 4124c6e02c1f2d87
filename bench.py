#!/usr/bin/env python3
"""bench.py -- reads/sec through the k-mer -> pileup hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|5]

One SAMPLE = one pass of the hot path over the reads of one sample, inputs already resident in HBM:
    bk_sample_begin  ->  bk_push_reads_packed_device per read batch (scan_count + level2 kernels)
    [-> RCCL reduce-scatter(sum) of the k-mer counter plane when one sample's reads are sharded over ranks]
    ->  bk_sample_finalize (KMC thresholds + map_kmers kernels).  Outputs stay in HBM.
One STEP = `samples_per_step` samples (config.samples_per_step; 32 for config 2, so that the K timed steps are >= 100 ms of
GPU work and the driver's sampler sees them).  Samples are independent (call.rs:212 handles a run's samples one after the
other): `--in-flight` engines on the same device tables (bk_engine_fork) take them in turn, each on its own stream.  The read
batches rotate over >= 8 distinct synthetic batches (> 256 MiB in all), so no batch is served from the Infinity Cache.
value = reads of all K steps / wall time of the K steps (barrier + synchronize on both sides, max over ranks).

--config 2 (default; BASELINE configs[1]): wuhan_ref k=21, samples of 1,000,000 x 150 bp single-end reads, seed 2.
           N > 1: every rank scans its own 1M-read shard of each sample (weak scaling); the counter plane is reduce-scattered
           once per sample, every rank maps its part, the small pileups are combined (max / sum).
--config 3 (configs[2]): the four golden SARS-CoV-2 genomes k=21, one sample = 10,000,000 pairs (2 x 150 bp) derived from
           ON765678.1, seed 3 (reference selection + pileup).  N > 1: the sample's 1M-pair batches are dealt to the ranks.
--config 5 (configs[4]): 100 synthetic strains k=31, 64 samples x 1,000,000 reads, whole samples per GPU (no collective).

--gpus N > 1 started with plain `python` spawns N ranks itself (before anything touches the GPU); under torchrun
(RANK / WORLD_SIZE in the environment) it is one of the ranks.  It fails loudly when N ranks cannot be had.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
ALGO_BYTES_PER_READ = 40    # 150 bases x 2 bit, padded to a 40 B record, read once
GOLDEN = os.path.join(ROOT, "tests", "golden", "4_sarscov2")
STRAINS4 = ["wuhan_ref.fasta", "OM223929.1.fasta", "ON765678.1.fasta", "PX392231.1.fasta"]   # tests/build_tests.rs:11-14


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5], help="BASELINE.json configs[] entry (1-based as in SURVEY.md §8d)")
    ap.add_argument("--reads", type=int, default=1000000, help="reads (config 3: pairs) per batch = per GPU per sample in config 2")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--samples-per-step", type=int, default=0, help="0 = the config's default (32 / 1 / 64)")
    ap.add_argument("--batches", type=int, default=0, help="distinct read batches resident in HBM (0 = the config's default: 8 / 10 / 64)")
    ap.add_argument("--strains", type=int, default=100, help="config 5: number of synthetic strains")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all host cores)")
    ap.add_argument("--cpu-sample", type=int, default=1000000, help="reads (pairs) per pass through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allreduce", action="store_true", help="N > 1: all-reduce the counter plane instead of reduce-scatter + sharded finalize")
    ap.add_argument("--wide", action="store_true", help="N > 1: move the counter plane as 64-bit integers even when 32 bits would do")
    ap.add_argument("--in-flight", type=int, default=3, help="samples in flight per GPU (bk_engine_fork: shared index tables, own counter "
                    "planes / outputs / stream); 1 = strictly one sample after the other")
    ap.add_argument("--selected-only", action="store_true", help="bk_params.pileup_selected_only: votes for the selected genome only (two finalize "
                    "passes); what `bronko call` runs with -- the default is the reference's literal map_kmers: every genome's rows")
    ap.add_argument("--dry-run", action="store_true", help="launcher check without a GPU: start the ranks, form the process group (use "
                    "--backend gloo), all-reduce the rank ids, print n_gpus / rccl_ranks and stop -- no hot path, no number")
    ap.add_argument("--backend", default="nccl", help="testing aid: 'gloo' lets several ranks share one GPU (rank r uses GPU r mod #GPUs)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes.  Nothing in this process has touched the
    GPU (torch.cuda.device_count() does not initialise it); the children are fresh interpreters, never an exec of this one."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev == 0 and not args.dry_run:
        raise SystemExit("bench.py: no GPU visible (there is no CPU fallback for the hot path)")
    if args.backend == "nccl" and n_dev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- refusing to report a %d-GPU number" % (args.gpus, n_dev, args.gpus))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:   # a rank failed: stop exactly the processes started here
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    sys.exit(rc)


def source_build_id():
    """sha256 over the kernel / engine sources the shipped libbronko_hip.so was built from (profiles/pmc_traffic.json carries the
    id it was captured with: a traffic figure of another build is not reported)."""
    h = hashlib.sha256()
    for rel in ("bronko_amd/csrc/bk_kernels.hip", "bronko_amd/csrc/bk_engine.cpp", "bronko_amd/csrc/bk_device.h", "bronko_amd/csrc/bk_kernels.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def main():
    args = parse_args()
    bad_env = sorted(k for k in os.environ if k.startswith("BK_"))
    if bad_env:
        raise SystemExit("bench.py: testing / measurement aids are set in the environment (%s); refusing to time anything" % ", ".join(bad_env))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)   # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np  # noqa: F401
    import torch
    if args.dry_run:
        import torch.distributed as dist
        n = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            t = torch.tensor([rank + 1], dtype=torch.int64)
            dist.all_reduce(t)
            if int(t.item()) != world * (world + 1) // 2:
                raise SystemExit("bench.py --dry-run: all-reduce over the ranks gave %d" % int(t.item()))
            n = dist.get_world_size()
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": n, "backend": args.backend, "value": None}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    from bronko_amd import Params, synth
    from bronko_amd.dist import ShardedFinalize, allreduce_counters
    from bronko_amd.hostlib import HostIndex

    cfg = args.config
    rl = args.read_len
    t_setup = time.perf_counter()
    # ---- index + engine --------------------------------------------------------------------------------------------------
    if cfg == 2:
        k, n_mates = 21, 1
        ref_paths = [os.path.join(GOLDEN, "wuhan_ref.fasta")]
        ix = HostIndex.build(k, ref_paths, threads=4)
        files = None
    elif cfg == 3:
        k, n_mates = 21, 2
        ref_paths = [os.path.join(GOLDEN, n) for n in STRAINS4]
        ix = HostIndex.build(k, ref_paths, threads=4)
        files = None
    else:
        k, n_mates = 31, 1
        files = synth.strain_files(synth.read_fasta_bytes(os.path.join(GOLDEN, "wuhan_ref.fasta")), args.strains)
        ix = HostIndex.build_mem(k, files, threads=min(32, os.cpu_count() or 4))
        ref_paths = None
    t_index = time.perf_counter()
    eng = ix.engine(Params(device=local_rank, pileup_selected_only=args.selected_only))
    t_engine = time.perf_counter()

    # ---- synthetic samples, generated on the GPU (bronko_amd.synth: the same splitmix64 streams as the numpy generator) ----
    # samples[i] = list of (mate, words tensor, lens tensor, n_records); every rank builds only what it will push
    samples = []
    keep = []          # codes of the first batch, for the CPU baseline (rank 0)
    if cfg == 2:
        # reference + 20 SNPs + 20 iSNVs, 0.5 % substitution errors.  Batch b of rank r: seed 2*1000003 + r
        # + 7919 b (batch 0 is round 1's batch).  A sample is one batch per rank = `world` shards of one sample.
        genome, isnv = synth.sample_genome(synth.read_fasta_bytes(ref_paths[0]), 2)
        nb = args.batches or 8
        for b in range(nb):
            codes = synth.single_end_codes_torch(genome, args.reads, rl, 2 * 1000003 + rank + 7919 * b, err=0.005, isnv=isnv, device=dev)
            w, l = synth.pack_codes_torch(codes)
            if b == 0 and rank == 0:
                keep = [codes[:args.cpu_sample].to(torch.uint8).cpu().numpy()]
            samples.append([(0, w, l, args.reads)])
            del codes
        sps = args.samples_per_step or 32
        reads_per_sample_rank = args.reads
        reads_per_sample_total = args.reads * world
        scaling = "weak"
        sharded_reads = world > 1
        workload = ("BASELINE configs[1]: SARS-CoV-2 single ref (wuhan_ref, 29903 bp), k=21, n_fixed=2, samples of %d synthetic %d bp "
                    "single-end reads per GPU, 0.5%% substitution errors, seed 2; %d distinct batches rotate" % (args.reads, rl, nb))
    elif cfg == 3:
        genome, isnv = synth.sample_genome(synth.read_fasta_bytes(ref_paths[2]), 3)
        nb = args.batches or 10
        pushes = []
        for b in range(nb):
            if b % world != rank:
                continue
            c1, c2 = synth.paired_codes_torch(genome, args.reads, rl, 3, err=0.005, isnv=isnv, device=dev, row0=b * args.reads)
            for m, c in enumerate((c1, c2)):
                w, l = synth.pack_codes_torch(c)
                pushes.append((m, w, l, args.reads))
            if b == 0 and rank == 0:
                keep = [c1[:args.cpu_sample].to(torch.uint8).cpu().numpy(), c2[:args.cpu_sample].to(torch.uint8).cpu().numpy()]
            del c1, c2
        samples.append(pushes)
        sps = args.samples_per_step or 1
        reads_per_sample_rank = sum(p[3] for p in pushes)
        reads_per_sample_total = 2 * nb * args.reads
        scaling = "strong"
        sharded_reads = world > 1
        workload = ("BASELINE configs[2]: 4 SARS-CoV-2 strains (wuhan_ref, OM223929.1, ON765678.1, PX392231.1), k=21, one sample = %d "
                    "synthetic pairs (2 x %d bp, fragment 300) derived from ON765678.1, seed 3, pushed as %d batches of %d pairs per mate; "
                    "reference selection + pileup" % (nb * args.reads, rl, nb, args.reads))
    else:
        nb = args.batches or 64
        for s in range(nb):
            if s % world != rank:
                continue
            src = s % args.strains
            genome, isnv = synth.sample_genome(files[src][1][0][1], 5 + s)
            codes = synth.single_end_codes_torch(genome, args.reads, rl, 5 * 1000003 + s, err=0.005, isnv=isnv, device=dev)
            w, l = synth.pack_codes_torch(codes)
            if s == 0 and rank == 0:
                keep = [codes[:args.cpu_sample].to(torch.uint8).cpu().numpy()]
            samples.append([(0, w, l, args.reads)])
            del codes
        sps = args.samples_per_step or len(samples)
        reads_per_sample_rank = args.reads
        reads_per_sample_total = args.reads
        scaling = "strong"
        sharded_reads = False
        workload = ("BASELINE configs[4]: %d synthetic strains (wuhan_ref + 300 substitutions each), k=31, %d samples x %d synthetic %d bp "
                    "single-end reads (sample s derived from strain s mod %d), whole samples per GPU" % (args.strains, nb, args.reads, rl, args.strains))
    torch.cuda.synchronize()
    t_data = time.perf_counter()
    resident = sum(w.numel() * 4 + l.numel() * 2 for smp in samples for (_, w, l, _) in smp)

    # ---- engines: samples in flight --------------------------------------------------------------------------------------
    # Each engine launches on its own HIP stream (created with the engine); torch sees it as an ExternalStream and every torch /
    # torch.distributed operation on the engine's buffers is enqueued on it: the collectives are ordered against the kernels by
    # the stream.
    n_fly = max(1, args.in_flight)
    engs = [eng] + [eng.fork() for _ in range(n_fly - 1)]
    streams = [torch.cuda.ExternalStream(e.stream_ptr(), device=dev) for e in engs]
    torch.cuda.set_stream(streams[0])
    counters = [[torch.as_tensor(_DevArray(e.counters_ptr(m), e.counter_len, "<i8"), device=dev) for m in range(n_mates)] for e in engs] if sharded_reads else None

    # one sample's reads sharded over ranks: reduce-scatter of the counter plane + each rank maps its part + max / sum of the small
    # pileups (include/bronko_hip.h); --allreduce selects the plain form (all-reduce the plane, every rank maps everything)
    sharded = sharded_reads and not args.allreduce
    if sharded and 64 % world != 0:
        raise SystemExit("bench.py: the sharded finalize needs a rank count that divides 64 (got %d); use --allreduce" % world)
    narrow = reads_per_sample_total * max(rl - k + 1, 1) < 2 ** 31 and not args.wide
    shard_fin = [ShardedFinalize(e, n_mates, rank, world, dev, narrow=narrow) for e in engs] if sharded else None

    def run_sample(i, j):
        e = engs[j]
        with torch.cuda.stream(streams[j]):
            e.sample_begin()
            for (m, w, l, n) in samples[i % len(samples)]:
                e.push_reads_device(m, w.data_ptr(), w.shape[1], l.data_ptr(), n)
            if sharded:
                shard_fin[j]()   # RCCL over xGMI: reduce-scatter(sum) + 3 small all-reduces
                return
            if sharded_reads:
                for m in range(n_mates):
                    e.counters_ptr(m)                    # (a plane nothing was pushed to is zeroed by this call)
                    allreduce_counters(counters[j][m])   # RCCL over xGMI: ONE all-reduce(sum) of the u64 k-mer occurrence counters
            e.sample_finalize(n_mates)

    state = {"i": 0}

    def step(pool):
        for _ in range(sps):
            i = state["i"]
            state["i"] = i + 1
            run_sample(i, i % pool)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timing(kind):
        for e in engs:
            e.timing_enable(kind)

    def timing_read():
        tot_ms, tot_n = [0.0] * 4, [0] * 4
        for e in engs:
            ms, n = e.timing_read(reset=True)
            tot_ms = [a + b for a, b in zip(tot_ms, ms)]
            tot_n = [a + b for a, b in zip(tot_n, n)]
        return tot_ms, tot_n

    for _ in range(args.warmup):
        step(len(engs))
    fence()
    # timed region: only the dominant kernel is bracketed by HIP events (on its launch stream, two records per launch)
    timing(2)
    timing_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(len(engs))
    fence()
    dt = time.perf_counter() - t0
    kms_fly, kn_fly = timing_read()
    last_engine = (state["i"] - 1) % len(engs)
    res = engs[last_engine].sample_download(n_mates, arrays=False)   # sanity: the last timed sample really produced its statistics
    # The same samples strictly one after the other on one engine, every kernel kind bracketed: a single sample's turnaround and
    # each kernel's own duration with nothing running next to it -- the figure the roofline object is about (in the timed
    # region above a scan shares the CUs with the other samples' kernels).
    n_serial = max(2, min(32 if reads_per_sample_rank <= 2000000 else 8, args.steps * sps))   # (32 short samples: the average of 8 moved by 10 % from run to run)
    for i in range(8):   # (untimed: the chip settles into running one sample at a time)
        run_sample(i, 0)
    fence()
    timing(1)
    timing_read()
    ts0 = time.perf_counter()
    for i in range(n_serial):
        run_sample(i, 0)
    fence()
    serial_ms = (time.perf_counter() - ts0) / n_serial * 1e3
    kms_solo, kn_solo = timing_read()
    timing(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_samples_timed = args.steps * sps
    if cfg == 5:
        total_reads = n_samples_timed * reads_per_sample_total * world   # every rank runs `sps` whole samples per step
    else:
        total_reads = n_samples_timed * reads_per_sample_total
    value = total_reads / dt

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure is the one measured
    # with rocprofv3 (separate --pmc passes) on this workload and build, committed under profiles/
    build_id = source_build_id()
    traffic = valu_insts = None
    traffic_note = "profiles/pmc_traffic.json absent"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tj.get("build_id") != build_id:
            traffic_note = "profiles/pmc_traffic.json was captured for build %s, this is %s: not reported" % (tj.get("build_id"), build_id)
        elif tj.get("config", 2) != cfg or tj.get("workload_reads") != args.reads or tj.get("read_len") != rl:
            traffic_note = "profiles/pmc_traffic.json is for another workload: not reported"
        else:
            traffic = tj["traffic_bytes_per_launch"]
            valu_insts = tj.get("valu_wave_insts_per_launch")
            traffic_note = "rocprofv3 --pmc passes of this build (%s)" % tj.get("captured", "profiles/")
    except (OSError, ValueError, KeyError):
        pass

    launches_per_sample = max(kn_solo[0] // max(n_serial, 1), 1)
    reads_per_launch = reads_per_sample_rank / launches_per_sample
    scan_ms = kms_solo[0] / max(kn_solo[0], 1)
    scan_ms_fly = kms_fly[0] / max(kn_fly[0], 1)
    algo_bytes = ALGO_BYTES_PER_READ * reads_per_launch
    achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    per_sample = lambda ms: ms / max(n_serial, 1)   # noqa: E731
    out = {
        "metric": "reads/sec through call k-mer->pileup, SARS-CoV-2 k=%d" % k,
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "rccl_ranks": dist.get_world_size() if world > 1 else 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_sample": dt / n_samples_timed * 1e3,
        "serial_ms_per_sample": serial_ms,   # one sample at a time on one engine (not the headline: see config.samples_in_flight)
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": workload, "baseline_config": cfg, "k": k, "read_len": rl, "samples_per_step": sps,
                   "reads_per_sample": reads_per_sample_total, "reads_per_gpu_per_sample": reads_per_sample_rank, "mates": n_mates,
                   "samples_in_flight": len(engs), "resident_input_bytes": resident,
                   "pileup_rows": "selected genome only (bk_params.pileup_selected_only)" if args.selected_only else "every genome (call.rs:1305-1384)",
                   "parallelism": ("single GPU" if world == 1 else
                                   "whole samples per GPU over %d GPUs, no collective" % world if not sharded_reads else
                                   ("one sample's reads sharded over %d GPUs; RCCL reduce-scatter(sum) of the k-mer counter plane%s, sharded finalize, "
                                    "all-reduce(max / sum) of the pileups" % (world, " as int32" if narrow else "")) if sharded else
                                   "one sample's reads sharded over %d GPUs; RCCL all-reduce(sum) of the k-mer counter plane" % world)},
        "roofline": {"bound": "hbm", "kernel": "scan_count_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                     "avg_kernel_ms": scan_ms, "launches": kn_solo[0],
                     "measured": "HIP events around the kernel on its launch stream, samples run one at a time after the timed region",
                     "avg_kernel_ms_in_flight": scan_ms_fly, "launches_in_flight": kn_fly[0],   # timed region: sharing the CUs
                     "reads_per_launch": reads_per_launch, "algorithmic_bytes_per_launch": algo_bytes,
                     # the kernel is VALU-issue bound, not HBM bound: wave-instructions per launch from the
                     # committed PMC pass; a wave64 VALU instruction occupies one of the chip's 1024 SIMDs for 4 cycles
                     "valu_wave_insts_per_launch": valu_insts,
                     "valu_busy_frac": (valu_insts * 4 / (1024 * 2.4e9) / (scan_ms * 1e-3)) if valu_insts and scan_ms > 0 else None},
        # per sample, one sample at a time (solo): what each kernel kind costs with nothing next to it
        "kernels_ms_per_sample_solo": {"scan_count": per_sample(kms_solo[0]), "finalize": per_sample(kms_solo[1]),
                                       "memset_copy": per_sample(kms_solo[2]), "level2": per_sample(kms_solo[3])},
        "check": {"perfect_kmers": [int(x) for x in res.stats.sum(axis=0)[:, 0][:8]], "variant_kmers": [int(x) for x in res.stats.sum(axis=0)[:, 1][:8]],
                  "kmers_scanned": [int(x) for x in res.kmer_stats[:, 1]]},
        "build": {"source_sha256_16": build_id},
        "setup_s": {"index_build": t_index - t_setup, "engine_create": t_engine - t_index, "synthetic_reads_on_gpu": t_data - t_engine},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline = the oracle (C restatement of bronko v0.1.0: exact strand-specific k-mer counting standing in for `kmc -t`,
        # then map_kmers over chunks in parallel as call.rs:1279-1281) on a bounded sample of the same workload, on this box's
        # host cores.  Checker code, timed here only as the reported baseline -- it is never on the product path.
        from oracle import oracle as orc
        ncpu = os.cpu_count() or 1
        threads = args.cpu_threads or ncpu
        oix = orc.Index.build_mem(k, files) if cfg == 5 else orc.Index.build(k, ref_paths)
        mates = [synth.BASES[c] for c in keep]
        n_s = len(mates[0])
        passes, cdt, s1, s2 = 0, 0.0, 0.0, 0.0
        while passes < 8 and (cdt < 4.0 or passes < 2):
            c0 = time.perf_counter()
            _, secs = orc.sample_pileup_mt(oix, mates, threads)
            cdt += time.perf_counter() - c0
            s1 += secs[0]
            s2 += secs[1]
            passes += 1
        n1 = min(n_s, 150000)
        c0 = time.perf_counter()
        orc.sample_pileup_mt(oix, [m[:n1] for m in mates], 1)
        one = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": n_s * len(mates) * passes / cdt, "unit": "reads/s", "cores": threads, "kind": "port",
                               "sample": "%d passes over the first %d %s of batch 0: CPU restatement of bronko v0.1.0 (not the upstream "
                                         "binary), %d threads of %d host cores, %.1f s in total"
                                         % (passes, n_s, "pairs" if len(mates) == 2 else "reads", threads, ncpu, cdt),
                               "stage_seconds_per_pass": {"count_all_kmers (kmc stand-in)": s1 / passes, "map_kmers": s2 / passes},
                               "single_thread_value": n1 * len(mates) / one, "single_thread_sample": "%d reads, 1 thread" % (n1 * len(mates))}
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
