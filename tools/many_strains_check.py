#!/usr/bin/env python3
"""BASELINE config 5 shape at reduced scale: N synthetic strains (wuhan_ref + 300 substitutions each), k = 31,
reads from one strain; HIP path vs oracle (parity) + timings of index build / engine creation / one sample.
usage: tools/many_strains_check.py [n_strains=30] [n_reads=200000] [--no-oracle]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth, _ffi  # noqa: E402
if os.environ.get("BK_L2_STATS"):
    _ffi.use_testing_library(True)
from bronko_amd.hostlib import HostIndex  # noqa: E402

n_strains = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
use_oracle = "--no-oracle" not in sys.argv
base = synth.read_fasta_bytes(os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta"))
files = synth.strain_files(base, n_strains)
t0 = time.time()
ix = HostIndex.build_mem(31, files, threads=8)
t1 = time.time()
print("index: %d strains, %d buckets, %d entries, built in %.1f s" % (n_strains, ix.n_buckets, ix.n_entries, t1 - t0), flush=True)
eng = ix.engine(Params())
t2 = time.time()
print("engine created in %.1f s: n_slots=%d counter plane=%.1f MB" % (t2 - t1, eng.n_slots, eng.counter_len * 8 / 1e6), flush=True)
src = 7 % n_strains
gm, isnv = synth.sample_genome(files[src][1][0][1], 5)
codes = synth.single_end_codes(gm, n_reads, 150, 55, isnv=isnv)
words, lens = synth.pack_codes(codes)
for rep in range(2):
    t3 = time.time()
    eng.sample_begin()
    eng.push_reads(0, words, lens)
    res = eng.sample_finish(1)
    t4 = time.time()
print("sample of %d reads: %.1f ms (incl. H2D + D2H)" % (n_reads, (t4 - t3) * 1e3), flush=True)
best = int(np.argmax(res.stats[0, :, 0] / 1.0))
print("perfect k-mers per strain (top 3):", sorted([(int(v), i) for i, v in enumerate(res.stats[0, :, 0])], reverse=True)[:3], "source strain", src)
if use_oracle:
    from oracle import oracle as orc
    t5 = time.time()
    oix = orc.Index.build_mem(31, files)
    pile = orc.sample_pileup(oix, [synth.codes_to_ascii(codes)])
    t6 = time.time()
    ok = all(np.array_equal(a, b) for a, b in zip(res.arrays(), pile.arrays())) and np.array_equal(res.stats, pile.stats)
    for name, a, b in zip(("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"), res.arrays(), pile.arrays()):
        bad = np.nonzero(a != b)[0]
        if len(bad):
            print(name, "differs in", len(bad), "cells; first", bad[:5], "hip", a[bad[:5]], "oracle", b[bad[:5]])
    if not np.array_equal(res.stats, pile.stats):
        d = np.nonzero((res.stats != pile.stats).any(axis=2))[1]
        print("stats differ for files", d[:10], res.stats[0, d[:3]], pile.stats[0, d[:3]])
    print("oracle (index + sample) %.1f s; parity: %s" % (t6 - t5, "OK" if ok else "MISMATCH"))
    sys.exit(0 if ok else 1)
