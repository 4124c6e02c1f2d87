#!/usr/bin/env python3
"""How long the host takes to ENQUEUE one config-2 sample (begin + push + finalize, no wait) against how long the GPU takes to run it:
whether bench.py's samples-in-flight figure is bound by the submitting thread.   gpurun -- python3 tools/host_submit_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bronko_amd import Params, synth
from bronko_amd.hostlib import HostIndex
path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_eng = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g, isnv = synth.sample_genome(synth.read_fasta_bytes(path), 2)
codes = synth.single_end_codes_torch(g, n, 150, 2000006, isnv=isnv, device=dev)
w, l = synth.pack_codes_torch(codes)
torch.cuda.synchronize()
ix = HostIndex.build(21, [path], threads=4)
eng = ix.engine(Params())
engs = [eng] + [eng.fork() for _ in range(n_eng - 1)]
def sample(e):
    e.sample_begin()
    e.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), n)
    e.sample_finalize(1)
for i in range(5 * n_eng): sample(engs[i % n_eng])
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for i in range(N): sample(engs[i % n_eng])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%d reads, %d engines: enqueue %.1f us per sample; all done after %.1f us per sample" % (n, n_eng, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
for e in engs[1:]: e.close()
eng.close()
