#!/usr/bin/env python3
"""Where bk_engine_create spends its time for a large index (testing build, BK_CREATE_TIMING=1: wall-clock of the host-side
table construction phases on stderr).  usage: tools/create_timing.py [n_strains 100] [k 31]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BK_CREATE_TIMING"] = "1"
from bronko_amd import Params, synth, _ffi
from bronko_amd.hostlib import HostIndex
_ffi.use_testing_library(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
files = synth.strain_files(synth.read_fasta_bytes(os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")), n)
t0 = time.time()
ix = HostIndex.build_mem(k, files, threads=min(32, os.cpu_count() or 4))
t1 = time.time()
eng = ix.engine(Params(pileup_selected_only=1))
t2 = time.time()
print("index build %.2f s, engine create %.2f s" % (t1 - t0, t2 - t1))
eng.close()
