#!/usr/bin/env python3
"""Scan kernel time against the number of reads in the launch (config 2 shape): the fixed part of a launch (LDS set-up, the
last tile's flush, the prefix sum and the slab write) against the part that grows with the reads.
usage: tools/scan_size_sweep.py [ablate | release]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bronko_amd import Params, synth, _ffi
from bronko_amd.hostlib import HostIndex
if len(sys.argv) > 1 and sys.argv[1] == "release":   # the product build (no switches)
    pass
else:
    _ffi.use_testing_library(True)
    if len(sys.argv) > 1:
        os.environ["BK_SCAN_ABLATE"] = sys.argv[1]
path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
dev = torch.device("cuda", 0)
g, isnv = synth.sample_genome(synth.read_fasta_bytes(path), 2)
N = 4000000
codes = synth.single_end_codes_torch(g, N, 150, 2000006, isnv=isnv, device=dev)
w, l = synth.pack_codes_torch(codes)
torch.cuda.synchronize()
ix = HostIndex.build(21, [path], threads=4)
eng = ix.engine(Params())
for n in (64, 1024, 16384, 65536, 262144, 524288, 1000000, 1048576, 2000000, 4000000):
    for rep in range(5):
        if rep == 1:
            eng.timing_enable(1); eng.timing_read(reset=True)
        eng.sample_begin()
        eng.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), n)
        eng.sample_finalize(1)
    ms, cnt = eng.timing_read(reset=True)
    print("%8d reads: scan %.4f ms (%d launches), level2+fold %.4f, finalize %.4f" % (n, ms[0] / max(cnt[0], 1), cnt[0] // 4, ms[3] / 4, ms[1] / 4), flush=True)
eng.close()
