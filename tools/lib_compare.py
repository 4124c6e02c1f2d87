#!/usr/bin/env python3
"""Kernel times of one config-2 sample through the release library and through its -DBK_TESTING twin (sanity: the twin must
differ by its aids only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bronko_amd import Params, synth, _ffi
from bronko_amd.hostlib import HostIndex
paths = [os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")]
dev = torch.device("cuda", 0)
g, isnv = synth.sample_genome(synth.read_fasta_bytes(paths[0]), 2)
c = synth.single_end_codes_torch(g, 1000000, 150, 2000006, isnv=isnv, device=dev)
w, l = synth.pack_codes_torch(c)
for mode in sys.argv[1:] or ["release", "testing", "sparse", "stats"]:
    _ffi.use_testing_library(mode != "release")
    os.environ.pop("BK_SPARSE_FINALIZE", None)
    if mode == "stats":
        os.environ["BK_L2_STATS"] = "1"
    if mode == "sparse":                       # the touch-list finalize of large indexes, forced onto this small one
        os.environ["BK_SPARSE_FINALIZE"] = "1"
    ix = HostIndex.build(21, paths, threads=4)
    eng = ix.engine(Params())
    for rep in range(3):
        if rep == 2:
            eng.timing_enable(1); eng.timing_read(reset=True)
        eng.sample_begin()
        eng.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), 1000000)
        eng.sample_finalize(1)
    ms, n = eng.timing_read(reset=True)
    res = eng.sample_download(1, arrays=False)
    print(mode, "scan %.3f finalize %.3f memset %.3f level2 %.3f ms; perfect %d variant %d" % (ms[0], ms[1], ms[2], ms[3], res.stats[0, 0, 0], res.stats[0, 0, 1]), flush=True)
    eng.close()
