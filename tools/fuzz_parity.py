#!/usr/bin/env python3
"""Randomised parity: small random indexes (repeats, reverse-complement repeats, homopolymers, several sequences and
files, k from 11 to 31, window variants) and read sets (lengths 20..400, 0..12 % substitutions, indels, foreign reads,
N symbols), HIP path vs oracle, ci = 1 so that every single k-mer occurrence shows.  Usage: fuzz_parity.py [iterations [seed0 [first iteration]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import Params, pack_reads, _ffi
_ffi.use_testing_library(True)   # BK_LDS_BINS / BK_REF_IN_LDS / BK_MAX_LAUNCH_RECORDS exist in the -DBK_TESTING build only
from tests import helpers
from oracle import oracle as orc

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = b"ACGT"
COMP = bytes.maketrans(b"ACGT", b"TGCA")

def rand_seq(rng, n):
    return bytes(B[i] for i in rng.integers(0, 4, n))

def make_genome(rng, n):
    g = bytearray(rand_seq(rng, n))
    for _ in range(int(rng.integers(0, 4))):          # direct and reverse-complement repeats
        ln = int(rng.integers(15, 120)); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln))
        seg = bytes(g[a:a + ln])
        if rng.random() < 0.5: seg = seg.translate(COMP)[::-1]
        g[b:b + ln] = seg
    if rng.random() < 0.4:                            # homopolymer / dinucleotide stretch
        ln = int(rng.integers(10, 60)); a = int(rng.integers(0, n - ln))
        unit = rand_seq(rng, int(rng.integers(1, 3)))
        g[a:a + ln] = (unit * ln)[:ln]
    return bytes(g)

def mutate(rng, g, n_sub):
    g = bytearray(g)
    for p in rng.integers(0, len(g), n_sub):
        g[p] = B[(B.index(g[p]) + int(rng.integers(1, 4))) & 3]
    return bytes(g)

bad = 0
t0 = time.time()
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for it in range(first, first + iters):
    rng = np.random.default_rng(seed0 * 100003 + it)
    k = int(rng.choice([11, 15, 19, 21, 21, 21, 25, 31, 31]))
    base = make_genome(rng, int(rng.integers(300, 6000)))
    files = []
    for f in range(int(rng.choice([1, 1, 2, 2, 3, 3, 5, 8]))):
        g = base if f == 0 else mutate(rng, base, int(rng.integers(0, 30)))
        cuts = sorted(set([0, len(g)] + [int(x) for x in rng.integers(0, len(g), int(rng.integers(0, 3)))]))
        seqs = [("s%d_%d" % (f, i), g[a:b]) for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])) if b - a >= 1]
        files.append(("file%d" % f, seqs))
    try:
        ix = orc.Index.build_mem(k, files)
    except Exception as e:                             # e.g. every sequence shorter than k
        continue
    n_fixed = int(rng.choice([2, 2, 0, 1, 5]))
    full = bool(rng.random() < 0.15)
    # how the engine is driven: exact k-mer statistics table, a small LDS window (cells outside it go through Level 2), the
    # reference read from global memory, pushes split into several launches
    stats = bool(rng.random() < 0.25)
    sel_only = bool(len(files) > 1 and rng.random() < 0.3)   # two finalize passes, votes for the selected genome only
    prm = Params(ci=int(rng.choice([1, 1, 1, 2, 3])), n_fixed=n_fixed, use_full_kmer=int(full), full_kmer_stats=stats, kmer_table_log2=21,
                 pileup_selected_only=sel_only)
    for var in ("BK_LDS_BINS", "BK_REF_IN_LDS", "BK_MAX_LAUNCH_RECORDS", "BK_SPARSE_FINALIZE"):
        os.environ.pop(var, None)
    if rng.random() < 0.4: os.environ["BK_SPARSE_FINALIZE"] = "1"          # touch lists instead of plane scans (large indexes)
    if rng.random() < 0.2: os.environ["BK_LDS_BINS"] = str(int(rng.integers(64, 2000)))
    if rng.random() < 0.1: os.environ["BK_REF_IN_LDS"] = "0"
    if rng.random() < 0.2: os.environ["BK_MAX_LAUNCH_RECORDS"] = str(int(rng.integers(1, 700)))
    eng = helpers.engine_from_oracle_index(ix, prm)
    src = mutate(rng, base, int(rng.integers(0, 12)))
    reads = []
    err = float(rng.choice([0.0, 0.005, 0.02, 0.05, 0.12]))
    # (round 6) one iteration in seven is a sample that is mostly NOT the reference's: Level 2 rolls reads marked whole, sixteen or
    # more to a batch of marked records, against the half-k-mer presence filters -- which must strike nothing that touches the index
    p_foreign = float(rng.uniform(0.5, 0.95)) if rng.random() < 0.15 else 0.05
    for _ in range(int(rng.integers(1, 3000))):
        ln = int(rng.integers(20, min(400, len(src))))
        a = int(rng.integers(0, len(src) - ln + 1))
        r = bytearray(src[a:a + ln])
        u = rng.random()
        if u < 0.05 or rng.random() < p_foreign - 0.05:                                 # foreign
            r = bytearray(rand_seq(rng, ln))
            if rng.random() < 0.3 and ln > 30:                                          # ... with a stretch of the reference in it (its k-mers do touch)
                q = int(rng.integers(0, ln - 25)); r[q:q + 25] = src[a + q:a + q + 25]
        elif u < 0.10 and ln > 60:                                                      # chimera
            b2 = int(rng.integers(0, len(src) - ln + 1)); r[ln // 2:] = src[b2 + ln // 2:b2 + ln]
        elif u < 0.15 and ln > 40:                                                      # deletion / insertion
            p = int(rng.integers(10, ln - 10))
            r = r[:p] + (bytearray(rand_seq(rng, int(rng.integers(1, 4)))) if rng.random() < 0.5 else bytearray()) + r[p + int(rng.integers(0, 4)):]
        for p in np.nonzero(rng.random(len(r)) < err)[0]:
            r[p] = B[(B.index(r[p]) + int(rng.integers(1, 4))) & 3] if r[p] in B else r[p]
        if rng.random() < 0.02 and len(r) > 5: r[int(rng.integers(0, len(r)))] = ord("N")
        r = bytes(r)
        if rng.random() < 0.5: r = r.translate(COMP)[::-1]
        reads.append(r)
    if os.environ.get("FUZZ_VERBOSE"): print("it=%d k=%d n_fixed=%d full=%d files=%d cells=%d reads=%d err=%.3f maxlen=%d" % (it, k, n_fixed, full, len(files), sum(len(s[1]) for f in files for s in f[1]), len(reads), err, max(len(r) for r in reads)), flush=True)
    mates = [reads]
    if rng.random() < 0.3 and len(reads) > 1:                                           # paired: two mate files
        h = len(reads) // 2
        mates = [reads[:h], reads[h:]]
    batch = int(rng.integers(1, 500)) if rng.random() < 0.3 else None
    ascii_path = bool(rng.random() < 0.3)
    res = helpers.hip_sample(eng, mates, k, batch=batch, ascii_path=ascii_path)
    if rng.random() < 0.3:                                                              # the engine is left clean: a second sample gives the same
        res = helpers.hip_sample(eng, mates, k, batch=batch, ascii_path=ascii_path)
    pile = orc.sample_pileup(ix, mates, n_fixed=n_fixed, use_full_kmer=full, ci=int(prm.ci))
    try:
        if sel_only:
            # statistics of every genome; rows of the selected genome only (the others stay zero)
            assert np.array_equal(res.stats, pile.stats) and np.array_equal(res.present, pile.present), ("stats", res.stats.tolist(), pile.stats.tolist())
            best = orc.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0))
            lo, ncell = ix.genome_cells(best) if best >= 0 else (0, 0)
            for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
                got, ref = getattr(res, name), getattr(pile, name)
                assert np.array_equal(got[lo * 4:(lo + ncell) * 4], ref[lo * 4:(lo + ncell) * 4]), (name, "selected genome", best)
                assert not got[:lo * 4].any() and not got[(lo + ncell) * 4:].any(), (name, "other genomes")
        else:
            helpers.assert_same_pileup(res, pile)
        assert res.kmer_stats[:, 1].tolist() == pile.kmc_stats[:, 1].tolist(), ("total k-mers", res.kmer_stats[:, 1], pile.kmc_stats[:, 1])
        if stats:
            assert res.kmer_stats[:, 2:4].tolist() == pile.kmc_stats[:, 2:4].tolist(), ("kmc stats", res.kmer_stats, pile.kmc_stats)
    except AssertionError as e:
        bad += 1
        print("MISMATCH it=%d seed=%d k=%d n_fixed=%d full=%d files=%d reads=%d err=%.3f mates=%d batch=%s ascii=%d stats=%d sel_only=%d env=%s: %s" %
              (it, seed0, k, n_fixed, full, len(files), len(reads), err, len(mates), batch, ascii_path, stats, sel_only,
               {v: os.environ[v] for v in ("BK_LDS_BINS", "BK_REF_IN_LDS", "BK_MAX_LAUNCH_RECORDS", "BK_SPARSE_FINALIZE") if v in os.environ}, str(e)[:300]), flush=True)
    eng.close(); ix.close()
print("%d iterations, %d mismatches, %.0f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
