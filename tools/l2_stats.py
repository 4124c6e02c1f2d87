#!/usr/bin/env python3
"""What does the scan leave to Level 2, and why?  (-DBK_TESTING build, BK_L2_STATS=1: tallies printed by bk_sample_finalize.)
usage: tools/l2_stats.py [config 2|3|5] [reads-or-pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BK_L2_STATS"] = "1"
import torch
from bronko_amd import Params, synth, _ffi
from bronko_amd.hostlib import HostIndex
_ffi.use_testing_library(True)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
names = ["wuhan_ref.fasta", "OM223929.1.fasta", "ON765678.1.fasta", "PX392231.1.fasta"]
paths = [os.path.join(ROOT, "tests", "golden", "4_sarscov2", x) for x in names]
dev = torch.device("cuda", 0)
prm = Params()
if cfg == 5:
    files = synth.strain_files(synth.read_fasta_bytes(paths[0]), 100)
    ix = HostIndex.build_mem(31, files, threads=min(32, os.cpu_count() or 4))
    g, isnv = synth.sample_genome(files[0][1][0][1], 5)
    mates = [synth.single_end_codes_torch(g, n, 150, 5 * 1000003, err=0.005, isnv=isnv, device=dev)]
    prm = Params(pileup_selected_only=1)
elif cfg == 2:
    ix = HostIndex.build(21, paths[:1], threads=4)
    g, isnv = synth.sample_genome(synth.read_fasta_bytes(paths[0]), 2)
    mates = [synth.single_end_codes_torch(g, n, 150, 2000006, isnv=isnv, device=dev)]
else:
    ix = HostIndex.build(21, paths, threads=4)
    g, isnv = synth.sample_genome(synth.read_fasta_bytes(paths[2]), 3)
    mates = list(synth.paired_codes_torch(g, n, 150, 3, isnv=isnv, device=dev))
eng = ix.engine(prm)
eng.sample_begin()
keep = []
for m, c in enumerate(mates):
    w, l = synth.pack_codes_torch(c)
    keep.append((w, l))
    torch.cuda.synchronize()   # the records are written on torch's stream, the engine reads them on its own
    eng.push_reads_device(m, w.data_ptr(), w.shape[1], l.data_ptr(), n)
eng.sample_finalize(len(mates))
res = eng.sample_download(len(mates), arrays=False)
print("config %d, %d reads per mate: perfect per genome %s" % (cfg, n, res.stats.sum(axis=0)[:, 0].tolist()))
