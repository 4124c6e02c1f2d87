#!/usr/bin/env python3
"""One config-2 sample at a time fed from sequence lines resident in HBM (K0 on the device): per-kind kernel times per sample."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth
from bronko_amd.hostlib import HostIndex
dev = torch.device("cuda", 0)
p = [os.path.join(ROOT, "tests/golden/4_sarscov2/wuhan_ref.fasta")]
g, isnv = synth.sample_genome(synth.read_fasta_bytes(p[0]), 2)
c = synth.single_end_codes_torch(g, 1000000, 150, 2000006, isnv=isnv, device=dev)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
bases = lut[c.long()].reshape(-1).contiguous(); offs = (torch.arange(1000001, dtype=torch.int64, device=dev) * 150).contiguous()
eng = HostIndex.build(21, p, threads=4).engine(Params())
for rep in range(6):
    if rep == 3: eng.timing_enable(1); eng.timing_read(reset=True)
    eng.sample_begin(); eng.push_reads_ascii_device(0, bases.data_ptr(), offs.data_ptr(), 1000000, 150000000, 150); eng.sample_finalize(1)
ms, n = eng.timing_read(reset=True)
print("per sample: scan %.4f finalize %.4f k0+zero %.4f level2 %.4f ms" % tuple(m / 3 for m in ms))
