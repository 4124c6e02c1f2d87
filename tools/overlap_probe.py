#!/usr/bin/env python3
"""Probe: samples alternated over E engines on E streams (sample i+1's scan overlaps sample i's finalize) vs one engine."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth
from bronko_amd.hostlib import HostIndex
n_eng = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
ref_path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
ref = synth.read_fasta_bytes(ref_path)
ix = HostIndex.build(21, [ref_path], threads=4)
genome, isnv = synth.sample_genome(ref, 2)
codes = synth.single_end_codes(genome, 1000000, 150, 2000006, err=0.005, isnv=isnv)
words, lens = synth.pack_codes(codes)
stride = words.shape[1]
d_words = torch.from_numpy(words.view(np.int32)).to(dev); d_lens = torch.from_numpy(lens.view(np.int16)).to(dev)
engs, streams = [], []
for i in range(n_eng):
    e = ix.engine(Params()); s = torch.cuda.Stream(device=dev); e.set_stream(s.cuda_stream); engs.append(e); streams.append(s)
def step(i):
    e = engs[i % n_eng]
    e.sample_begin(); e.push_reads_device(0, d_words.data_ptr(), stride, d_lens.data_ptr(), len(lens)); e.sample_finalize(1)
for i in range(6): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("engines %d: %.3f ms/sample, %.2f G reads/s" % (n_eng, dt / steps * 1e3, 1e6 * steps / dt / 1e9))
