import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, synth, pack_reads
from bronko_amd.hostlib import HostIndex
from oracle import oracle as orc
n_strains = int(sys.argv[1]); n_reads = int(sys.argv[2])
base = synth.read_fasta_bytes(os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta"))
files = []
for s in range(n_strains):
    g = np.frombuffer(base, np.uint8).copy()
    r = synth.splitmix64(5000 + s, 600)
    pos = (r[0::2] % np.uint64(len(g))).astype(np.int64)
    sh = (r[1::2] % np.uint64(3)).astype(np.int64) + 1
    for p, d in zip(pos, sh):
        g[p] = synth.BASES[(int(synth.CODE[g[p]]) + int(d)) & 3]
    files.append(("strain%03d" % s, [("seq%03d" % s, g.tobytes())]))
ix = HostIndex.build_mem(31, files, threads=8)
eng = ix.engine(Params())
oix = orc.Index.build_mem(31, files)
gm, isnv = synth.sample_genome(files[7 % n_strains][1][0][1], 5)
codes = synth.single_end_codes(gm, n_reads, 150, 55, isnv=isnv)
reads = synth.codes_to_ascii(codes)
def run(rs):
    w, l = pack_reads(rs, 31)
    eng.sample_begin(); eng.push_reads(0, w, l); res = eng.sample_finish(1)
    pile = orc.sample_pileup(oix, [rs])
    ok = all(np.array_equal(a, b) for a, b in zip(res.arrays(), pile.arrays())) and np.array_equal(res.stats, pile.stats)
    return (not ok), res, pile
cur = reads
b, res, pile = run(cur)
print("full bad:", b, flush=True)
n = 2
while len(cur) > 3 and b:
    chunk = max(1, len(cur) // n)
    reduced = False
    for i in range(0, len(cur), chunk):
        cand = cur[:i] + cur[i + chunk:]
        if cand and run(cand)[0]:
            cur = cand; n = max(n - 1, 2); reduced = True; break
    if not reduced:
        if chunk == 1: break
        n = min(len(cur), n * 2)
print("minimal reads:", len(cur), flush=True)
b, res, pile = run(cur)
d = np.nonzero((res.stats != pile.stats).any(axis=2))
print("stats hip", res.stats[0][d[1][:4]].tolist(), "oracle", pile.stats[0][d[1][:4]].tolist(), "files", d[1][:8])
km, ct, st = orc.count_kmers(31, cur)
bad_files = set(int(x) for x in d[1])
for kmer, c in zip(km, ct):
    p = orc.Pileup(oix); orc.map_kmers(oix, [kmer], [c], p)
    if p.stats.sum():
        s = "".join("ACGT"[(int(kmer) >> (2 * (30 - i))) & 3] for i in range(31))
        single = [s.encode()] * int(c)
        b1, r1, p1 = run(single)
        if b1:
            v, rc = orc.canonical_kmer(s)
            nz = np.nonzero((r1.stats != p1.stats).any(axis=2))[1]
            print("BAD kmer", s, "canon %x rc=%d" % (v, rc), "count", int(c), "files", nz[:8], "hip", r1.stats[0][nz[:3]].tolist(), "oracle", p1.stats[0][nz[:3]].tolist(),
                  "votes hip", int(r1.fwd_nk.sum() + r1.rev_nk.sum()), "oracle", int(p1.fwd_nk.sum() + p1.rev_nk.sum()), flush=True)
if len(cur) <= 20:
    for r in cur: print(r.decode())
