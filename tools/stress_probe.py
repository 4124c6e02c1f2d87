#!/usr/bin/env python3
"""Throughput away from the benchmark's sweet spot: higher error rates, longer reads, reads that are not from the reference."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth
from bronko_amd.hostlib import HostIndex
ref_path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
ref = synth.read_fasta_bytes(ref_path)
ix = HostIndex.build(21, [ref_path], threads=4)
eng = ix.engine(Params())
dev = torch.device("cuda", 0)
gm, isnv = synth.sample_genome(ref, 2)
def run(tag, codes):
    w, l = synth.pack_codes(codes)
    dw = torch.from_numpy(w.view(np.int32)).to(dev); dl = torch.from_numpy(l.view(np.int16)).to(dev)
    def step():
        eng.sample_begin(); eng.push_reads_device(0, dw.data_ptr(), w.shape[1], dl.data_ptr(), len(l)); eng.sample_finalize(1)
    for _ in range(3): step()
    torch.cuda.synchronize()
    eng.timing_enable(1); eng.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    ms, n = eng.timing_read(reset=True); eng.timing_enable(0)
    print("%-34s %7.3f ms/sample  %6.2f G bases/s   scan %.3f finalize %.3f" % (tag, dt * 1e3, codes.size / dt / 1e9, ms[0] / 10, ms[1] / 10), flush=True)
N = 500000
run("150 bp, 0.5 % errors", synth.single_end_codes(gm, N, 150, 5, err=0.005, isnv=isnv))
run("150 bp, 2 % errors", synth.single_end_codes(gm, N, 150, 5, err=0.02, isnv=isnv))
run("150 bp, 5 % errors", synth.single_end_codes(gm, N, 150, 5, err=0.05, isnv=isnv))
run("250 bp, 0.5 % errors", synth.single_end_codes(gm, N * 150 // 250, 250, 5, err=0.005, isnv=isnv))
run("1000 bp, 1 % errors", synth.single_end_codes(gm, N * 150 // 1000, 1000, 5, err=0.01, isnv=isnv))
rnd = (synth.splitmix64(99, N * 150) & np.uint64(3)).astype(np.uint8).reshape(N, 150)
run("150 bp, random (not the reference)", rnd)
mix = synth.single_end_codes(gm, N, 150, 5, err=0.005, isnv=isnv); mix[::2] = rnd[::2]
run("150 bp, half random", mix)
