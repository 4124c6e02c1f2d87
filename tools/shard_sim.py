#!/usr/bin/env python3
"""Debug aid: the sharded finalize simulated with `world` engines on one GPU vs the plain finalize (stats only)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth
from bronko_amd.dist import DeviceVector
from bronko_amd.hostlib import HostIndex
world = int(sys.argv[1]); n_reads = int(sys.argv[2])
ref_path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
ref = synth.read_fasta_bytes(ref_path)
ix = HostIndex.build(21, [ref_path], threads=4)
genome, isnv = synth.sample_genome(ref, 2)
codes = synth.single_end_codes(genome, n_reads, 150, 7, err=0.005, isnv=isnv)
words, lens = synth.pack_codes(codes)
one = ix.engine(Params())
one.sample_begin(); one.push_reads(0, words, lens); full = one.sample_finish(1)
print("plain  ", full.stats[0].tolist(), int(full.fwd_nk.sum()), int(full.fwd_depth.sum()))
engs = [ix.engine(Params()) for _ in range(world)]
for r, e in enumerate(engs):
    e.sample_begin()
    lo, hi = len(lens) * r // world, len(lens) * (r + 1) // world
    e.push_reads(0, words[lo:hi], lens[lo:hi])
torch.cuda.synchronize()
n = engs[0].counter_len; part = n // world
planes = [torch.as_tensor(DeviceVector(e.counters_ptr(0), n), device="cuda:0") for e in engs]
total = planes[0].clone()
for p in planes[1:]: total += p
for r, p in enumerate(planes):
    p.copy_(total)
torch.cuda.synchronize()
cells4 = engs[0].total_cells * 4
piles, sums = [], []
for r, e in enumerate(engs):
    e.sample_finalize_shard(1, r, world)
    piles.append(torch.as_tensor(DeviceVector(e.pileup_ptr(), 4 * cells4), device="cuda:0"))
    sp, sn = e.shard_sums(); sums.append(torch.as_tensor(DeviceVector(sp, sn), device="cuda:0"))
torch.cuda.synchronize()
for r in range(world): print("rank", r, sums[r][:3].tolist())
piles[0][:2 * cells4] = torch.stack([p[:2 * cells4] for p in piles]).max(dim=0).values
piles[0][2 * cells4:] = torch.stack([p[2 * cells4:] for p in piles]).sum(dim=0)
sums[0].copy_(torch.stack(sums).sum(dim=0))
torch.cuda.synchronize()
engs[0].sample_merge_shards()
res = engs[0].sample_download(1)
print("sharded", res.stats[0].tolist(), int(res.fwd_nk.sum()), int(res.fwd_depth.sum()))
print("equal:", all(np.array_equal(a, b) for a, b in zip(res.arrays(), full.arrays())) and np.array_equal(res.stats, full.stats))
