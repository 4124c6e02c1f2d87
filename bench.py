#!/usr/bin/env python3
"""bench.py -- reads/sec through the k-mer -> pileup hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

One SAMPLE = one pass of the hot path over the reads of one sample, inputs already resident in HBM:
    bk_sample_begin  ->  bk_push_reads_packed_device per read batch (scan_items + bin_count + nbatch + level2 kernels)
    [-> RCCL reduce-scatter(sum) of the k-mer counter plane when one sample's reads are sharded over ranks]
    ->  bk_sample_finalize (KMC thresholds + map_kmers kernels).  Outputs stay in HBM.
One STEP = `samples_per_step` samples (config.samples_per_step; 512 for config 2, so that the K = 20 timed steps are >= 1 s of
GPU work and the driver's sampler sees them; 288 until the path passed 6 G reads/s).  Samples are independent (call.rs:212
handles a run's samples one after the other): `--in-flight` engines on the same device tables (bk_engine_fork) take them in
turn, each on its own stream.  The read batches rotate over >= 8 distinct synthetic batches (> 256 MiB in all), so no batch is served from the Infinity Cache.
value = reads of all K steps / wall time of the K steps (barrier + synchronize on both sides, max over ranks).

--config 2 (default; BASELINE configs[1]): wuhan_ref k=21, samples of 1,000,000 x 150 bp single-end reads, seed 2.
           N > 1: every rank scans its own 1M-read shard of each sample (weak scaling); the counter plane is reduce-scattered
           once per sample, every rank maps its part, the small pileups are combined (max / sum).
--config 3 (configs[2]): the four golden SARS-CoV-2 genomes k=21, one sample = 10,000,000 pairs (2 x 150 bp) derived from
           ON765678.1, seed 3 (reference selection + pileup).  N > 1: the sample's 1M-pair batches are dealt to the ranks.
--config 4 (configs[3]): wuhan_ref k=21, ONE sample of 200 x 1,000,000 reads, seed 4; rank r pushes the batches b = r (mod N);
           N > 1: one reduce-scatter of the counter plane per sample, sharded finalize.  N = 1 keeps the 8.4 GB of records resident.
--config 5 (configs[4]): 100 synthetic strains k=31, 64 samples x 1,000,000 reads, whole samples per GPU (no collective).
The default run (config 2, one GPU) also measures configs 3 and 5 (literal and selected-only) on a few samples each and reports
them under "other_configs", and feeds the config-2 steps once more from device-resident ASCII through K0 ("value_with_k0").

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

# HIP gives a process four hardware queues by default and maps its streams onto them: with four or more engines in flight (a stream
# each) two streams share a queue and their kernels wait for each other.  Eight queues, set before the runtime initialises (the
# `bronko` binary does the same in main()); an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[] entry (1-based as in SURVEY.md §8d)")
    ap.add_argument("--reads", type=int, default=1000000, help="reads (config 3: pairs) per batch = per GPU per sample in config 2")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--samples-per-step", type=int, default=0, help="0 = the config's default (512 / 1 / 1 / 64)")
    ap.add_argument("--batches", type=int, default=0, help="distinct read batches resident in HBM (0 = the config's default: 8 / 10 / 200 / 64)")
    ap.add_argument("--strains", type=int, default=100, help="config 5: number of synthetic strains")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all host cores)")
    ap.add_argument("--cpu-sample", type=int, default=1000000, help="reads (pairs) per pass through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-events-every", type=int, default=16, help="timed region: HIP events around every N-th launch of the dominant kernel (an event record "
                    "idles its stream for ~10 us on the GPU: bracketing every launch slowed down what it measured; max 255)")
    ap.add_argument("--from-ascii", action="store_true", help="config 2 on one GPU: feed the timed steps from sequence lines resident in HBM through K0 (what "
                    "\"value_with_k0\" of the default line measures briefly); for the rocprofv3 captures of K0 -- not the headline configuration")
    ap.add_argument("--experiment", action="store_true", help="tools/*.sh sweeps: BK_* testing variables may be set (the line says \"experiment\": true and is no result)")
    ap.add_argument("--allreduce", action="store_true", help="N > 1: all-reduce the counter plane instead of reduce-scatter + sharded finalize")
    ap.add_argument("--width", default="auto", choices=["auto", "16", "32", "64"], help="N > 1: bits per counter on the wire (bk_shard_transport). auto = "
                    "measured during the warm-up (bk_shard_measure), then fixed for the timed region; a width that is too narrow is detected, never silent")
    ap.add_argument("--no-other-configs", action="store_true", help="config 2 on one GPU: skip the bounded measurements of configs 3 and 5 and the K0 figure")
    ap.add_argument("--in-flight", type=int, default=0, help="0 = the config's default (4; config 4, whose one sample is 200 scan launches: 3).  ""samples in flight per GPU (bk_engine_fork: shared index tables, own counter "
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
    csrc = os.path.join(ROOT, "bronko_amd", "csrc")
    names = []
    with open(os.path.join(csrc, "Makefile")) as f:   # every file of the Makefile's SRC and HDR lists: what the library is built from
        for line in f:
            if line.startswith(("SRC :=", "HDR :=")):
                names += line.split(":=", 1)[1].split()
    if not names:
        raise SystemExit("bench.py: no SRC / HDR lists in bronko_amd/csrc/Makefile")
    for rel in sorted(names):
        with open(os.path.normpath(os.path.join(csrc, rel)), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class Workload:
    """Index, engine and device-resident synthetic samples of one BASELINE config on this rank (built once; the bounded
    "other_configs" legs reuse it: config 5's selected-only run forks the literal run's engine with other parameters)."""

    def __init__(self, cfg, args, rank, world, local_rank, dev, reads, batches, selected_only=False, keep_ascii=False):
        import torch
        from bronko_amd import Params, synth
        from bronko_amd.hostlib import HostIndex
        self.cfg, self.rank, self.world = cfg, rank, world
        rl = self.rl = args.read_len
        t_setup = time.perf_counter()
        self.files = None
        if cfg in (2, 4):
            self.k, self.n_mates = 21, 1
            self.ref_paths = [os.path.join(GOLDEN, "wuhan_ref.fasta")]
            self.ix = HostIndex.build(self.k, self.ref_paths, threads=4)
        elif cfg == 3:
            self.k, self.n_mates = 21, 2
            self.ref_paths = [os.path.join(GOLDEN, n) for n in STRAINS4]
            self.ix = HostIndex.build(self.k, self.ref_paths, threads=4)
        else:
            self.k, self.n_mates = 31, 1
            self.files = synth.strain_files(synth.read_fasta_bytes(os.path.join(GOLDEN, "wuhan_ref.fasta")), args.strains)
            # build_indexes on the device (bk_build_index: what `bronko build` / `bronko call -g` run when a GPU is visible)
            from bronko_amd.engine import Engine, build_index_device
            self.ix = None
            built = build_index_device(self.k, self.files, device=local_rank)
            self.ref_paths = None
        t_index = time.perf_counter()
        if self.ix is None:
            self.eng = Engine(self.k, built[0], built[1], built[2], self.files, Params(device=local_rank, pileup_selected_only=selected_only))
            del built
        else:
            self.eng = self.ix.engine(Params(device=local_rank, pileup_selected_only=selected_only))
        self.selected_only = selected_only
        t_engine = time.perf_counter()

        # ---- synthetic samples, generated on the GPU (bronko_amd.synth: the same splitmix64 streams as the numpy generator) ----
        # samples[i] = list of (mate, words tensor, lens tensor, n_records); every rank builds only what it will push
        self.samples, self.keep, self.ascii = [], [], []
        k = self.k
        if cfg == 2:
            # reference + 20 SNPs + 20 iSNVs, 0.5 % substitution errors.  Batch b of rank r: seed 2*1000003 + r
            # + 7919 b (batch 0 is round 1's batch).  A sample is one batch per rank = `world` shards of one sample.
            genome, isnv = synth.sample_genome(synth.read_fasta_bytes(self.ref_paths[0]), 2)
            nb = batches or 8
            for b in range(nb):
                codes = synth.single_end_codes_torch(genome, reads, rl, 2 * 1000003 + rank + 7919 * b, err=0.005, isnv=isnv, device=dev)
                w, l = synth.pack_codes_torch(codes)
                if b == 0 and rank == 0:
                    self.keep = [codes[:args.cpu_sample].to(torch.uint8).cpu().numpy()]
                if keep_ascii:   # the same reads as sequence lines resident in HBM: what K0 (pack_reads_kernel) starts from
                    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
                    self.ascii.append((lut[codes.long()].reshape(-1).contiguous(),
                                       (torch.arange(reads + 1, dtype=torch.int64, device=dev) * rl).contiguous()))
                self.samples.append([(0, w, l, reads)])
                del codes
            self.sps = args.samples_per_step or 512
            self.reads_per_sample_rank = reads
            self.reads_per_sample_total = reads * world
            self.scaling = "weak"
            self.sharded_reads = world > 1
            self.workload = ("BASELINE configs[1]: SARS-CoV-2 single ref (wuhan_ref, 29903 bp), k=21, n_fixed=2, samples of %d synthetic %d bp "
                             "single-end reads per GPU, 0.5%% substitution errors, seed 2; %d distinct batches rotate" % (reads, rl, nb))
        elif cfg == 3:
            genome, isnv = synth.sample_genome(synth.read_fasta_bytes(self.ref_paths[2]), 3)
            nb = batches or 10
            pushes = []
            for b in range(nb):
                if b % world != rank:
                    continue
                c1, c2 = synth.paired_codes_torch(genome, reads, rl, 3, err=0.005, isnv=isnv, device=dev, row0=b * reads)
                for m, c in enumerate((c1, c2)):
                    w, l = synth.pack_codes_torch(c)
                    pushes.append((m, w, l, reads))
                if b == 0 and rank == 0:
                    self.keep = [c1[:args.cpu_sample].to(torch.uint8).cpu().numpy(), c2[:args.cpu_sample].to(torch.uint8).cpu().numpy()]
                del c1, c2
            self.samples.append(pushes)
            self.sps = args.samples_per_step or 1
            self.reads_per_sample_rank = sum(p[3] for p in pushes)
            self.reads_per_sample_total = 2 * nb * reads
            self.scaling = "strong"
            self.sharded_reads = world > 1
            self.workload = ("BASELINE configs[2]: 4 SARS-CoV-2 strains (wuhan_ref, OM223929.1, ON765678.1, PX392231.1), k=21, one sample = %d "
                             "synthetic pairs (2 x %d bp, fragment 300) derived from ON765678.1, seed 3, pushed as %d batches of %d pairs per mate; "
                             "reference selection + pileup" % (nb * reads, rl, nb, reads))
        elif cfg == 4:
            # as config 2 but ONE sample of 200 M reads, seed 4, in 1 M-read batches dealt round-robin to the
            # ranks (batch b: seed 4*1000003 + 7919 b).  At N = 1 all 200 batches (8.4 GB of records) stay resident.
            genome, isnv = synth.sample_genome(synth.read_fasta_bytes(self.ref_paths[0]), 4)
            nb = batches or 200
            pushes = []
            for b in range(nb):
                if b % world != rank:
                    continue
                codes = synth.single_end_codes_torch(genome, reads, rl, 4 * 1000003 + 7919 * b, err=0.005, isnv=isnv, device=dev)
                w, l = synth.pack_codes_torch(codes)
                if b == 0 and rank == 0:
                    self.keep = [codes[:args.cpu_sample].to(torch.uint8).cpu().numpy()]
                pushes.append((0, w, l, reads))
                del codes
            self.samples.append(pushes)
            self.sps = args.samples_per_step or 1
            self.reads_per_sample_rank = sum(p[3] for p in pushes)
            self.reads_per_sample_total = nb * reads
            self.scaling = "strong"
            self.sharded_reads = world > 1
            self.workload = ("BASELINE configs[3]: SARS-CoV-2 single ref (wuhan_ref), k=21, ONE sample of %d synthetic %d bp single-end reads, "
                             "0.5%% substitution errors, seed 4, pushed as %d batches of %d reads dealt round-robin to the ranks"
                             % (nb * reads, rl, nb, reads))
        else:
            nb = batches or 64
            for s in range(nb):
                if s % world != rank:
                    continue
                src = s % args.strains
                genome, isnv = synth.sample_genome(self.files[src][1][0][1], 5 + s)
                codes = synth.single_end_codes_torch(genome, reads, rl, 5 * 1000003 + s, err=0.005, isnv=isnv, device=dev)
                w, l = synth.pack_codes_torch(codes)
                if s == 0 and rank == 0:
                    self.keep = [codes[:args.cpu_sample].to(torch.uint8).cpu().numpy()]
                self.samples.append([(0, w, l, reads)])
                del codes
            self.sps = args.samples_per_step or len(self.samples)
            self.reads_per_sample_rank = reads
            self.reads_per_sample_total = reads
            self.scaling = "strong"
            self.sharded_reads = False
            self.workload = ("BASELINE configs[4]: %d synthetic strains (wuhan_ref + 300 substitutions each), k=31, %d samples x %d synthetic %d bp "
                             "single-end reads (sample s derived from strain s mod %d), whole samples per GPU" % (args.strains, nb, reads, rl, args.strains))
        torch.cuda.synchronize()
        t_data = time.perf_counter()
        self.reads, self.n_batches = reads, nb
        self.resident = sum(w.numel() * 4 + l.numel() * 2 for smp in self.samples for (_, w, l, _) in smp)
        self.setup_s = {"index_build": t_index - t_setup, "engine_create": t_engine - t_index, "synthetic_reads_on_gpu": t_data - t_engine}


def measure(wl, args, dev, dist, steps, warmup, sps=None, eng=None, selected_only=None, bounded=False, from_ascii=False):
    """Time `steps` steps of the workload (after `warmup`): the dict of the JSON line, without the CPU baseline.  bounded: the
    short form used for the "other_configs" legs (no one-at-a-time phase beyond 3 samples)."""
    import torch
    from bronko_amd.dist import ShardedFinalize, allreduce_counters
    rank, world, cfg, rl, k, n_mates = wl.rank, wl.world, wl.cfg, wl.rl, wl.k, wl.n_mates
    eng = eng or wl.eng
    selected_only = wl.selected_only if selected_only is None else selected_only
    sps = sps or wl.sps
    samples = wl.samples
    sharded_reads = wl.sharded_reads

    # ---- engines: samples in flight --------------------------------------------------------------------------------------
    # Each engine launches on its own HIP stream (created with the engine); torch sees it as an ExternalStream and every torch /
    # torch.distributed operation on the engine's buffers is enqueued on it: the collectives are ordered against the kernels by
    # the stream.
    n_fly = args.in_flight if args.in_flight > 0 else (3 if cfg == 4 else 4)
    engs = [eng] + [eng.fork() for _ in range(n_fly - 1)]
    streams = [torch.cuda.ExternalStream(e.stream_ptr(), device=dev) for e in engs]
    torch.cuda.set_stream(streams[0])
    sharded = sharded_reads and not args.allreduce
    counters = [[torch.as_tensor(_DevArray(e.counters_ptr(m), e.counter_len, "<i8"), device=dev) for m in range(n_mates)] for e in engs] \
        if sharded_reads and not sharded else None

    # one sample's reads sharded over ranks: the engine packs the counter plane for the wire, ONE reduce-scatter per mate file, each
    # rank maps its part, max / sum of the small pileups (include/bronko_hip.h); --allreduce selects the plain form (all-reduce the
    # plane, every rank maps everything)
    if sharded and 64 % world != 0:
        raise SystemExit("bench.py: the sharded finalize needs a rank count that divides 64 (got %d); use --allreduce" % world)
    width = "auto" if args.width == "auto" else int(args.width)
    # every engine in flight gets a process group of its own -- its own RCCL communicator and stream -- so that the reduce-scatter of
    # sample i does not queue behind the collectives of the samples in flight next to it (one default group serialises them on its
    # internal stream) and sample i + 1's scan runs under it
    groups = [dist.new_group(ranks=list(range(world))) for _ in engs] if sharded_reads else None
    shard_fin = [ShardedFinalize(e, n_mates, rank, world, dev, width=width, time_comm=True, group=groups[j]) for j, e in enumerate(engs)] if sharded else None

    def run_sample(i, j):
        e = engs[j]
        with torch.cuda.stream(streams[j]):
            e.sample_begin()
            if from_ascii:
                bases, offs = wl.ascii[i % len(wl.ascii)]
                e.push_reads_ascii_device(0, bases.data_ptr(), offs.data_ptr(), wl.reads, wl.reads * rl, rl)
            else:
                for (m, w, l, n) in samples[i % len(samples)]:
                    e.push_reads_device(m, w.data_ptr(), w.shape[1], l.data_ptr(), n)
            if sharded:
                shard_fin[j]()   # RCCL over xGMI: reduce-scatter(sum) + 3 small all-reduces
                return
            if sharded_reads:
                for m in range(n_mates):
                    e.counters_ptr(m)                    # (a plane nothing was pushed to is zeroed by this call)
                    allreduce_counters(counters[j][m], group=groups[j])   # RCCL over xGMI: ONE all-reduce(sum) of the u64 k-mer occurrence counters
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

    for _ in range(warmup):
        step(len(engs))
    # every engine in flight has run at least two samples before anything is timed: a fork's first sample sizes its per-launch
    # buffers (hipMalloc synchronises the device) -- `warmup` steps of one sample warmed one engine of three
    while state["i"] < 2 * len(engs):
        run_sample(state["i"], state["i"] % len(engs))
        state["i"] += 1
    fence()
    widths = None
    if sharded:
        # the width the warm-up's samples needed ("auto" measured every plane); the timed region runs at that fixed width -- no
        # measuring pass, no host synchronisation per sample -- and a counter that does not fit is detected on the device
        widths = sorted({w for f in shard_fin for w in f.last_widths}) or [64]
        if width == "auto":
            w = torch.tensor([max(widths)], dtype=torch.int64, device=dev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            for f in shard_fin:
                f.width = int(w.item())
        for f in shard_fin:
            f.comm_ms()
    # timed region: only the dominant kernel is bracketed by HIP events (on its launch stream, two records per launch)
    timing(2 | (args.timed_events_every << 8))
    timing_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(len(engs))
    fence()
    dt = time.perf_counter() - t0
    kms_fly, kn_fly = timing_read()
    comm = None
    if sharded:
        rs_ms = cb_ms = 0.0
        n_ev = 0
        for f in shard_fin:
            a, b, n = f.comm_ms()
            rs_ms, cb_ms, n_ev = rs_ms + a, cb_ms + b, n_ev + n
        over = any(e.transport_overflow() for e in engs)
        flag = torch.tensor([1 if over else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            raise SystemExit("bench.py: a counter did not fit the %d-bit transport in the timed region; rerun with a wider --width" % shard_fin[0].width)
        comm = {"width_bits": shard_fin[0].width, "plane_bytes": int(eng.counter_len * 8), "bytes_on_the_wire_per_rank_per_sample": int(shard_fin[0].bytes_sent),
                "reduce_scatter_ms_per_sample": rs_ms / max(n_ev, 1), "combine_ms_per_sample": cb_ms / max(n_ev, 1),
                "comm_ms_per_sample": (rs_ms + cb_ms) / max(n_ev, 1),
                "measured": "HIP events on the engine's stream around the reduce-scatter(s) and around the three small all-reduces; includes waiting for the slowest rank"}
    last_engine = (state["i"] - 1) % len(engs)
    res = engs[last_engine].sample_download(n_mates, arrays=False)   # sanity: the last timed sample really produced its statistics
    if comm:
        # How much of the collectives' time shows in the wall clock: the same steps once more with every collective replaced by its
        # local stand-in (each rank keeps its own part: wrong results, the same kernels) -- exposed = what the real steps took more.
        for f in shard_fin:
            f.local_only = True
        n_lo = max(2, steps // 4)
        step(len(engs))
        fence()
        tl = time.perf_counter()
        for _ in range(n_lo):
            step(len(engs))
        fence()
        dl = time.perf_counter() - tl
        if world > 1:
            t = torch.tensor([dl], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dl = float(t.item())
        for f in shard_fin:
            f.local_only = False
            f.comm_ms()
        any(e.transport_overflow() for e in engs)   # (the stand-in steps' flags are not the timed region's)
        no_comm = dl / (n_lo * sps) * 1e3
        comm["ms_per_sample_without_collectives"] = no_comm
        comm["comm_exposed_ms_per_sample"] = max(dt / (steps * sps) * 1e3 - no_comm, 0.0)
        comm["comm_hidden_ms_per_sample"] = max(comm["comm_ms_per_sample"] - comm["comm_exposed_ms_per_sample"], 0.0)
    # The same samples strictly one after the other on one engine (on the whole chip), every kernel kind bracketed: a single
    # sample's turnaround and each kernel's own duration with nothing running next to it -- the figure the roofline object is
    # about (in the timed region a scan shares the CUs with the other samples' kernels).
    n_serial = 3 if bounded else max(2, min(32 if wl.reads_per_sample_rank <= 2000000 else 8, steps * sps))   # (32 short samples: the average of 8 moved by 10 % from run to run)
    # (the forks are closed first: an engine with siblings leaves a quarter of the CUs to them -- alone it scans on the whole chip)
    n_in_flight = len(engs)
    fence()
    for e in engs[1:]:
        e.close()
    del engs[1:]
    for i in range(3 if bounded else 8):   # (untimed: the chip settles into running one sample at a time)
        run_sample(i, 0)
    fence()
    # ... first with no event in the stream (an event record idles the stream for ~10 us on the GPU: with every kernel kind
    # bracketed a sample took 0.04 ms longer than it does), then once more with the brackets for the kernels' own durations
    timing(0)
    ts0 = time.perf_counter()
    for i in range(n_serial):
        run_sample(i, 0)
    fence()
    serial_ms = (time.perf_counter() - ts0) / n_serial * 1e3
    timing(1)
    timing_read()
    for i in range(n_serial):
        run_sample(i, 0)
    fence()
    kms_solo, kn_solo = timing_read()
    timing(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_samples_timed = steps * sps
    if cfg == 5:
        total_reads = n_samples_timed * wl.reads_per_sample_total * world   # every rank runs `sps` whole samples per step
    else:
        total_reads = n_samples_timed * wl.reads_per_sample_total
    value = total_reads / dt

    launches_per_sample = max(kn_solo[0] // max(n_serial, 1), 1)
    reads_per_launch = wl.reads_per_sample_rank / launches_per_sample
    scan_ms = kms_solo[0] / max(kn_solo[0], 1)
    scan_ms_fly = kms_fly[0] / max(kn_fly[0], 1)
    algo_bytes = ALGO_BYTES_PER_READ * reads_per_launch
    achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    per_sample = lambda ms: ms / max(n_serial, 1)   # noqa: E731
    # which kernel the leg's roofline object is about: the longest single kernel of a sample run alone, from this run's own HIP-event
    # buckets.  Bucket 0 is one kernel (the scan), bucket 3 is Level 2 (level2_kernel; nbatch_kernel's share is small); bucket 1 is the
    # whole finalize -- several kernels, so it names a kernel only where one of them is known to dominate it
    scan_name = "scan_count_kernel" if cfg == 5 else "scan_items_kernel"
    l2_ms = kms_solo[3] / max(kn_solo[3], 1)
    dom_name, dom_ms, dom_launches = scan_name, scan_ms, kn_solo[0]
    if l2_ms > scan_ms:
        dom_name, dom_ms, dom_launches = "level2_kernel", l2_ms, kn_solo[3]
    dom_achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    out = {
        **({"experiment": True, "testing_env": sorted(k_ for k_ in os.environ if k_.startswith("BK_"))} if args.experiment else {}),
        "metric": "reads/sec through call k-mer->pileup, SARS-CoV-2 k=%d" % k,
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "rccl_ranks": dist.get_world_size() if world > 1 else 1,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "ms_per_sample": dt / n_samples_timed * 1e3,
        "serial_ms_per_sample": serial_ms,   # one sample at a time on one engine (not the headline: see config.samples_in_flight)
        "higher_is_better": True,
        "scaling": wl.scaling,
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": wl.workload, "baseline_config": cfg, "k": k, "read_len": rl, "samples_per_step": sps,
                   "reads_per_sample": wl.reads_per_sample_total, "reads_per_gpu_per_sample": wl.reads_per_sample_rank, "mates": n_mates,
                   "samples_in_flight": n_in_flight, "resident_input_bytes": wl.resident,
                   "input": "sequence lines (ASCII) resident in HBM -> K0 pack_reads_kernel -> records" if from_ascii else "2-bit packed records resident in HBM",
                   "pileup_rows": "selected genome only (bk_params.pileup_selected_only)" if selected_only else "every genome (call.rs:1305-1384)",
                   "parallelism": ("single GPU" if world == 1 else
                                   "whole samples per GPU over %d GPUs, no collective" % world if not sharded_reads else
                                   ("one sample's reads sharded over %d GPUs; RCCL reduce-scatter(sum) of the k-mer counter plane packed to %d-bit "
                                    "elements by the engine, sharded finalize, all-reduce(max / sum) of the pileups" % (world, shard_fin[0].width)) if sharded else
                                   "one sample's reads sharded over %d GPUs; RCCL all-reduce(sum) of the k-mer counter plane" % world)},
        # (config 5's index keeps its planes sparse and takes the whole-window scan_count_kernel; everything else the binned scan)
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": dom_achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": dom_achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                     "avg_kernel_ms": dom_ms, "launches": dom_launches,
                     "scan_kernel": {"kernel": scan_name, "avg_kernel_ms": scan_ms, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS},
                     "measured": "HIP events around the kernel on its launch stream, samples run one at a time on the whole chip after the timed region",
                     # timed region: from the record before a scan launch to the record behind it on its stream -- the kernel sharing
                     # the CUs with the other samples' kernels AND whatever time it waited for CUs: not a kernel duration
                     "avg_ms_in_flight_incl_queueing": scan_ms_fly, "launches_in_flight": kn_fly[0], "launches_in_flight_bracketed_every": args.timed_events_every,
                     "reads_per_launch": reads_per_launch, "algorithmic_bytes_per_launch": algo_bytes},
        # per sample, one sample at a time (solo): what each kernel kind costs with nothing next to it
        "kernels_ms_per_sample_solo": {"scan_count": per_sample(kms_solo[0]), "finalize": per_sample(kms_solo[1]),
                                       "memset_copy": per_sample(kms_solo[2]), "level2": per_sample(kms_solo[3])},
        "check": {"perfect_kmers": [int(x) for x in res.stats.sum(axis=0)[:, 0][:8]], "variant_kmers": [int(x) for x in res.stats.sum(axis=0)[:, 1][:8]],
                  "kmers_scanned": [int(x) for x in res.kmer_stats[:, 1]]},
    }
    if comm:
        out["comm"] = comm
    torch.cuda.set_stream(torch.cuda.default_stream(dev))   # (the engines' streams die with them)
    for e in reversed(engs[1:]):
        e.close()
    return out


def cpu_baseline(wl, args):
    """CPU baseline = the oracle (C restatement of bronko v0.1.0: exact strand-specific k-mer counting standing in for `kmc -t`,
    then map_kmers over chunks in parallel as call.rs:1279-1281) on a bounded sample of the same workload, on this box's host
    cores.  Checker code, timed here only as the reported baseline -- it is never on the product path."""
    from bronko_amd import synth
    from oracle import oracle as orc
    ncpu = os.cpu_count() or 1
    threads = args.cpu_threads or ncpu
    oix = orc.Index.build_mem(wl.k, wl.files) if wl.cfg == 5 else orc.Index.build(wl.k, wl.ref_paths)
    mates = [synth.BASES[c] for c in wl.keep]
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
    return {"value": n_s * len(mates) * passes / cdt, "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": "%d passes over the first %d %s of batch 0: CPU restatement of bronko v0.1.0 (not the upstream "
                      "binary), %d threads of %d host cores, %.1f s in total"
                      % (passes, n_s, "pairs" if len(mates) == 2 else "reads", threads, ncpu, cdt),
            "stage_seconds_per_pass": {"count_all_kmers (kmc stand-in)": s1 / passes, "map_kmers": s2 / passes},
            "single_thread_value": n1 * len(mates) / one, "single_thread_sample": "%d reads, 1 thread" % (n1 * len(mates))}


def pmc_traffic(cfg, reads, rl, build_id):
    """HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure is the one measured
    with rocprofv3 (separate --pmc passes) on this workload and build, committed under profiles/."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tj.get("build_id") != build_id:
            return None, None, "profiles/pmc_traffic.json was captured for build %s, this is %s: not reported" % (tj.get("build_id"), build_id)
        if tj.get("config", 2) != cfg or tj.get("workload_reads") != reads or tj.get("read_len") != rl:
            return None, None, "profiles/pmc_traffic.json is for another workload: not reported"
        return tj["traffic_bytes_per_launch"], tj.get("valu_wave_insts_per_launch"), "rocprofv3 --pmc passes of this build (%s)" % tj.get("captured", "profiles/")
    except (OSError, ValueError, KeyError):
        return None, None, "profiles/pmc_traffic.json absent"


def main():
    args = parse_args()
    bad_env = sorted(k for k in os.environ if k.startswith("BK_"))
    if bad_env and not args.experiment:
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

    from bronko_amd import Params

    cfg = args.config
    extras = cfg == 2 and world == 1 and not args.no_other_configs and not args.selected_only
    if args.from_ascii and (cfg != 2 or world != 1):
        raise SystemExit("bench.py: --from-ascii is for config 2 on one GPU")
    wl = Workload(cfg, args, rank, world, local_rank, dev, args.reads, args.batches, selected_only=args.selected_only, keep_ascii=extras or args.from_ascii)
    out = measure(wl, args, dev, dist, args.steps, args.warmup, from_ascii=args.from_ascii)
    build_id = source_build_id()
    traffic, valu_insts, note = pmc_traffic(cfg, args.reads, args.read_len, build_id)
    rf = out["roofline"]
    rf["traffic"], rf["traffic_source"] = traffic, note
    # the kernel is VALU-issue bound, not HBM bound: wave-instructions per launch from the committed PMC
    # pass; a wave64 VALU instruction occupies one of the chip's 1024 SIMDs for 4 cycles
    rf["valu_wave_insts_per_launch"] = valu_insts
    rf["valu_busy_frac"] = (valu_insts * 4 / (1024 * 2.4e9) / (rf["avg_kernel_ms"] * 1e-3)) if valu_insts and rf["avg_kernel_ms"] > 0 else None
    out["build"] = {"source_sha256_16": build_id}
    out["setup_s"] = wl.setup_s

    if extras:
        # SURVEY.md §8(d) words the metric as K0..K2: the same steps once more, fed from sequence lines resident in HBM through
        # pack_reads_kernel (no PCIe).  Algorithmic bytes of that figure: 150 B of ASCII read + 40 B of records written and read.
        k0 = measure(wl, args, dev, dist, max(2, args.steps // 4), 1, from_ascii=True, bounded=True)
        out["value_with_k0"] = {"value": k0["value"], "unit": "reads/s", "ms_per_sample": k0["ms_per_sample"], "serial_ms_per_sample": k0["serial_ms_per_sample"],
                                "input": k0["config"]["input"], "algorithmic_bytes_per_read": args.read_len + ALGO_BYTES_PER_READ, "check": k0["check"],
                                "k0_ms_per_sample_solo": k0["kernels_ms_per_sample_solo"]["memset_copy"]}
        wl.ascii = []
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, args)
    elif rank == 0:
        out["cpu_baseline"] = None

    if extras:
        # BASELINE configs 3 and 5 in the line the driver records: a few timed samples each after one warm-up sample (bounded: the
        # whole default run stays below ~90 s); the full-length figures come from `bench.py --config 3 / 5`.
        def brief(o, cpu=None):
            r = o["roofline"]
            d = {"value": o["value"], "unit": "reads/s", "ms_per_sample": o["ms_per_sample"], "serial_ms_per_sample": o["serial_ms_per_sample"],
                 "samples_timed": o["steps"] * o["config"]["samples_per_step"], "reads_per_sample": o["config"]["reads_per_sample"],
                 "workload": o["config"]["workload"], "pileup_rows": o["config"]["pileup_rows"], "check": o["check"],
                 "roofline": {"kernel": r["kernel"], "avg_kernel_ms": r["avg_kernel_ms"], "reads_per_launch": r["reads_per_launch"], "achieved_GBps": r["achieved"],
                              "frac": r["frac"], "scan_kernel": r["scan_kernel"]},
                 "kernels_ms_per_sample_solo": o["kernels_ms_per_sample_solo"]}
            if cpu:
                d["cpu_baseline"] = cpu
            return d
        others = {}
        wl.eng.close()
        del wl
        torch.cuda.empty_cache()
        a3 = argparse.Namespace(**vars(args))
        a3.samples_per_step, a3.cpu_sample = 1, 200000
        w3 = Workload(3, a3, rank, world, local_rank, dev, 1000000, 10)
        others["3"] = brief(measure(w3, a3, dev, dist, 4, 1, bounded=True))
        w3.eng.close()
        del w3
        torch.cuda.empty_cache()
        a5 = argparse.Namespace(**vars(args))
        a5.samples_per_step = 1
        w5 = Workload(5, a5, rank, world, local_rank, dev, 1000000, 6)
        others["5_literal"] = brief(measure(w5, a5, dev, dist, 6, 1, bounded=True))
        sel = w5.eng.fork(Params(device=local_rank, pileup_selected_only=True))
        others["5_selected_only"] = brief(measure(w5, a5, dev, dist, 9, 2, eng=sel, selected_only=True, bounded=True))
        sel.close()
        others["5_literal"]["setup_s"] = w5.setup_s
        out["other_configs"] = others
        w5.eng.close()
    else:
        wl.eng.close()

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
