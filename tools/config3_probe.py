#!/usr/bin/env python3
"""BASELINE config 3 shape at 1 M pairs: the four golden SARS-CoV-2 genomes (k = 21), paired-end reads derived from
ON765678.1, inputs resident on the device; per-sample time of begin + 2 pushes + finalize, kernel breakdown, selection."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth
from bronko_amd.hostlib import HostIndex
names = ["wuhan_ref.fasta", "OM223929.1.fasta", "ON765678.1.fasta", "PX392231.1.fasta"]
paths = [os.path.join(ROOT, "tests", "golden", "4_sarscov2", n) for n in names]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ix = HostIndex.build(21, paths, threads=4)
eng = ix.engine(Params())
gm, isnv = synth.sample_genome(synth.read_fasta_bytes(paths[2]), 3)
c1, c2 = synth.paired_codes(gm, n_pairs, 150, 3, isnv=isnv)
dev = torch.device("cuda", 0)
bufs = []
for c in (c1, c2):
    w, l = synth.pack_codes(c)
    bufs.append((torch.from_numpy(w.view(np.int32)).to(dev), torch.from_numpy(l.view(np.int16)).to(dev), w.shape[1], len(l)))
def step():
    eng.sample_begin()
    for m, (dw, dl, stride, n) in enumerate(bufs):
        eng.push_reads_device(m, dw.data_ptr(), stride, dl.data_ptr(), n)
    eng.sample_finalize(2)
for _ in range(3): step()
torch.cuda.synchronize()
eng.timing_enable(1); eng.timing_read(reset=True)
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
ms, n = eng.timing_read(reset=True)
res = eng.sample_download(2, arrays=False)
print("4 strains, %d pairs: %.3f ms per sample = %.2f G reads/s; kernels ms per sample: scan %.3f finalize %.3f fold %.3f memset %.3f"
      % (n_pairs, dt * 1e3, 2 * n_pairs / dt / 1e9, ms[0] / 10, ms[1] / 10, ms[3] / 10, ms[2] / 10))
print("perfect per genome:", res.stats.sum(axis=0)[:, 0].tolist())
