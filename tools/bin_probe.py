#!/usr/bin/env python3
"""bin_count_kernel / scan_items_kernel on reads WITHOUT true variants (only sequencing errors) vs the benchmark's sample (20 SNPs + 20
iSNVs): what the hot V bins of a sample's variants cost.  usage (GPU box): rocprofv3 --kernel-trace --stats ... -- python3 tools/bin_probe.py [plain|sample]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bronko_amd import Params, synth
from bronko_amd.hostlib import HostIndex
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
paths = [os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")]
dev = torch.device("cuda", 0)
g = synth.read_fasta_bytes(paths[0])
isnv = ()
if mode == "sample":
    g, isnv = synth.sample_genome(g, 2)
c = synth.single_end_codes_torch(g, 1000000, 150, 2000006, isnv=isnv, device=dev)
w, l = synth.pack_codes_torch(c)
ix = HostIndex.build(21, paths, threads=4)
eng = ix.engine(Params())
for rep in range(4):
    eng.sample_begin()
    eng.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), 1000000)
    eng.sample_finalize(1)
res = eng.sample_download(1, arrays=False)
print(mode, "perfect %d variant %d" % (res.stats[0, 0, 0], res.stats[0, 0, 1]))
eng.close()
