#!/usr/bin/env python3
"""Scan kernel time under the ablation switches of the -DBK_TESTING build (BK_SCAN_ABLATE: 2 = no V atomics, 5 = every N run to
level2_kernel, 6 = no mismatch loop, 10 = no table of hot V counters, 11 = no seed table, 7 = seeds only, 9 = not even the seeds, 8 = without prefix sum and slab, 4 = without Level 2)
12 = without cell_nat) for config 2, 3 or 5 shapes.  usage: tools/scan_ablate.py [2|3|5] [switches, comma separated] [config 5: sample number]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bronko_amd import Params, synth, _ffi
from bronko_amd.hostlib import HostIndex
RELEASE = len(sys.argv) > 2 and sys.argv[2] == "release"   # the product build, no switch (what tools/pmc_any.sh ... release profiles)
if not RELEASE:
    _ffi.use_testing_library(True)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 1000000
names = ["wuhan_ref.fasta", "OM223929.1.fasta", "ON765678.1.fasta", "PX392231.1.fasta"]
paths = [os.path.join(ROOT, "tests", "golden", "4_sarscov2", x) for x in names]
dev = torch.device("cuda", 0)
k = 21
if cfg == 5:
    k = 31
    smp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    files = synth.strain_files(synth.read_fasta_bytes(paths[0]), 100)
    g, isnv = synth.sample_genome(files[smp % 100][1][0][1], 5 + smp)
    mates = [synth.single_end_codes_torch(g, n, 150, 5 * 1000003 + smp, err=0.005, isnv=isnv, device=dev)]
    hix = HostIndex.build_mem(k, files, threads=32)
elif cfg == 2:
    paths = paths[:1]
    g, isnv = synth.sample_genome(synth.read_fasta_bytes(paths[0]), 2)
    mates = [synth.single_end_codes_torch(g, n, 150, 2000006, isnv=isnv, device=dev)]
else:
    g, isnv = synth.sample_genome(synth.read_fasta_bytes(paths[2]), 3)
    mates = list(synth.paired_codes_torch(g, n, 150, 3, isnv=isnv, device=dev))
packed = [synth.pack_codes_torch(c) for c in mates]
torch.cuda.synchronize()
for ab in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "11", "10", "2", "6", "7", "9")):
    if not RELEASE:
        os.environ["BK_SCAN_ABLATE"] = ab
    ix = hix if cfg == 5 else HostIndex.build(21, paths, threads=4)
    eng = ix.engine(Params(pileup_selected_only=(cfg == 5)))
    for rep in range(4):
        if rep == 1:
            eng.timing_enable(1); eng.timing_read(reset=True)
        eng.sample_begin()
        for m, (w, l) in enumerate(packed):
            eng.push_reads_device(m, w.data_ptr(), w.shape[1], l.data_ptr(), n)
        eng.sample_finalize(len(packed))
    ms, cnt = eng.timing_read(reset=True)
    print("config %d ablate %s: scan %.3f ms per launch, level2+fold %.3f, finalize %.3f per sample" % (cfg, ab, ms[0] / max(cnt[0], 1), ms[3] / max(cnt[3], 1), ms[1] / 3), flush=True)
    eng.close()
