#!/usr/bin/env python3
"""Reference selection + baseline noise + variant calls on the device (bk_sample_call) as a THROUGHPUT: config 2's
samples (1 M reads vs SARS-CoV-2, k = 21) scanned, finalized and called, one at a time and with four engines in flight -- the noise walk
of one sequence is a serial chain (one workgroup), the samples in flight run theirs side by side.
usage: call_throughput.py [samples, default 64] [in flight, default 4]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_fly = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-other-configs"]
import bench
import torch
args = bench.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
wl = bench.Workload(2, args, 0, 1, 0, dev, args.reads, args.batches)
engs = [wl.eng] + [wl.eng.fork() for _ in range(n_fly - 1)]
streams = [torch.cuda.ExternalStream(e.stream_ptr(), device=dev) for e in engs]

def run(i, j, call):
    e = engs[j]
    with torch.cuda.stream(streams[j]):
        e.sample_begin()
        for (m, w, l, n) in wl.samples[i % len(wl.samples)]:
            e.push_reads_device(m, w.data_ptr(), w.shape[1], l.data_ptr(), n)
        e.sample_finalize(wl.n_mates)
        if call:
            e.sample_call(wl.n_mates)

def timed(pool, call, n):
    for i in range(2 * pool):
        run(i, i % pool, call)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        run(i, i % pool, call)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

rows = []
for pool in (1, n_fly):
    a = timed(pool, False, n_samples)
    b = timed(pool, True, n_samples)
    rows.append((pool, a, b))
    print("%d in flight: k-mer -> pileup %.3f ms per sample; + selection, noise, calls %.3f ms per sample (the caller: %.3f ms)" % (pool, a, b, b - a))
summ, recs = engs[0].download_calls()
print("last sample of engine 0: genome %d, %d records (%d major, %d minor)" % (summ.file_id, summ.n_records, summ.n_major, summ.n_minor))
