#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: reads handed over as HOST buffers (never bench.py's `value`).

  ascii  : bk_push_reads_ascii  (150 B/read over PCIe, packed on the GPU, asynchronous 3-slot ring)
  packed : bk_push_reads_packed (42 B/read over PCIe, packed by the caller beforehand, synchronous copy)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth  # noqa: E402
from bronko_amd.hostlib import HostIndex  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ref_path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
ix = HostIndex.build(21, [ref_path])
eng = ix.engine(Params())
ref = synth.read_fasta_bytes(ref_path)
gm, isnv = synth.sample_genome(ref, 2)
codes = synth.single_end_codes(gm, N, 150, 2000006, isnv=isnv)
words, lens = synth.pack_codes(codes)
flat = np.ascontiguousarray(synth.BASES[codes].reshape(-1))
off = (np.arange(N + 1, dtype=np.uint64) * np.uint64(150))
B = 1 << 18
L = eng._L
for name in ("ascii", "packed"):
    for rep in range(3):
        t0 = time.perf_counter()
        eng.sample_begin()
        for i in range(0, N, B):
            j = min(N, i + B)
            if name == "ascii":
                o = np.ascontiguousarray(off[i:j + 1])
                rc = L.bk_push_reads_ascii(eng.h, 0, flat.ctypes.data, o.ctypes.data, j - i)
                assert rc == 0
            else:
                eng.push_reads(0, words[i:j], lens[i:j])
        eng.sample_finalize(1)
        res = eng.sample_download(1, arrays=False)
        dt = time.perf_counter() - t0
    print("%-6s %d reads in %.2f ms -> %.1f M reads/s (perfect=%d)" % (name, N, dt * 1e3, N / dt / 1e6, res.stats[0, 0, 0]))
