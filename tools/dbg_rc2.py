#!/usr/bin/env python3
"""BK_L2_STATS tallies for error-free reads at random starts in [lo, hi): both strands mixed, one strand only, sorted by start."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bronko_amd import Params, synth, _ffi
from bronko_amd.hostlib import HostIndex
_ffi.use_testing_library(True)
path = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
dev = torch.device("cuda", 0)
g = synth._t_genome(synth.read_fasta_bytes(path), dev)
L = g.numel()
lo, hi = int(sys.argv[1]) % L, (int(sys.argv[2]) - 1) % L + 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
ix = HostIndex.build(21, [path], threads=4)
eng = ix.engine(Params())
gen = torch.Generator(device="cpu"); gen.manual_seed(5)
start = torch.randint(lo, hi - 150 + 1, (n,), generator=gen).to(dev)
rev = (torch.randint(0, 2, (n,), generator=gen) != 0).to(dev)
def run(name, start, rev):
    codes = g[start[:, None] + torch.arange(150, device=dev)[None, :]]
    codes[rev] = (3 - codes[rev]).flip(1)
    w, l = synth.pack_codes_torch(codes.contiguous())
    torch.cuda.synchronize()   # (the engine reads on its own stream)
    print("case %s" % name, file=sys.stderr, flush=True)
    eng.sample_begin()
    eng.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), start.numel())
    eng.sample_finalize(1)
order = sys.argv[4] if len(sys.argv) > 4 else "mfr"
o = torch.argsort(start)
for ch in order:
    if ch == "m": run("mixed", start, rev)
    if ch == "f": run("fwd only", start, torch.zeros_like(rev))
    if ch == "r": run("rev only", start, torch.ones_like(rev))
torch.cuda.synchronize()
if "s" in order:
    run("sorted", start[o], rev[o])
    run("first 6400", start[:6400], rev[:6400])
    run("strand alternating, one start", torch.full((6400,), lo + 17, device=dev), (torch.arange(6400, device=dev) & 1) != 0)
eng.close()
