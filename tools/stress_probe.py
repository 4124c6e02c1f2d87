#!/usr/bin/env python3
"""Throughput away from the benchmark's sweet spot: higher error rates, longer reads, reads that are not from the reference.
usage: stress_probe.py [testing|release [rows [engines in flight, default 1]]]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth, _ffi
from bronko_amd.hostlib import HostIndex
if len(sys.argv) > 1 and sys.argv[1] == "testing":   # the -DBK_TESTING build: BK_SCAN_ABLATE and friends are honoured
    _ffi.use_testing_library(True)
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None   # run only the rows whose tag contains one of these
N_FLY = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ref_path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
ref = synth.read_fasta_bytes(ref_path)
ix = HostIndex.build(21, [ref_path], threads=4)
eng = ix.engine(Params())
dev = torch.device("cuda", 0)
engs = [eng] + [eng.fork() for _ in range(N_FLY - 1)]
streams = [torch.cuda.ExternalStream(e.stream_ptr(), device=dev) for e in engs]
gm, isnv = synth.sample_genome(ref, 2)
def run(tag, codes):
    if ONLY and not any(o in tag for o in ONLY):
        return
    dw, dl = synth.pack_codes_torch(codes)
    torch.cuda.synchronize()
    def step(j=0):
        e = engs[j]
        with torch.cuda.stream(streams[j]):
            e.sample_begin(); e.push_reads_device(0, dw.data_ptr(), dw.shape[1], dl.data_ptr(), dl.numel()); e.sample_finalize(1)
    for i in range(3 * N_FLY): step(i % N_FLY)
    torch.cuda.synchronize()
    if N_FLY > 1:
        t0 = time.perf_counter()
        for i in range(10 * N_FLY): step(i % N_FLY)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (10 * N_FLY)
        print("%-34s %7.3f ms/sample with %d in flight" % (tag, dt * 1e3, N_FLY), flush=True)
        return
    eng.timing_enable(1); eng.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    ms, n = eng.timing_read(reset=True); eng.timing_enable(0)
    print("%-34s %7.3f ms/sample  %6.2f G bases/s   scan %.3f finalize %.3f level2 %.3f" % (tag, dt * 1e3, codes.numel() / dt / 1e9, ms[0] / 10, ms[1] / 10, ms[3] / 10), flush=True)
N = 500000
run("150 bp, 0.5 % errors", synth.single_end_codes_torch(gm, N, 150, 5, err=0.005, isnv=isnv, device=dev))
run("150 bp, 2 % errors", synth.single_end_codes_torch(gm, N, 150, 5, err=0.02, isnv=isnv, device=dev))
run("150 bp, 5 % errors", synth.single_end_codes_torch(gm, N, 150, 5, err=0.05, isnv=isnv, device=dev))
run("250 bp, 0.5 % errors", synth.single_end_codes_torch(gm, N * 150 // 250, 250, 5, err=0.005, isnv=isnv, device=dev))
run("1000 bp, 1 % errors", synth.single_end_codes_torch(gm, N * 150 // 1000, 1000, 5, err=0.01, isnv=isnv, device=dev))
rnd = (synth._t_splitmix64(99, N * 150, dev) & 3).reshape(N, 150)
run("150 bp, random (not the reference)", rnd)
mix = synth.single_end_codes_torch(gm, N, 150, 5, err=0.005, isnv=isnv, device=dev); mix[::2] = rnd[::2]
run("150 bp, half random", mix)
