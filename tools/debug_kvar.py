"""debugging aid: HPV16 reads at k = 31 (or argv[1]) against the oracle with ci = 1, minimised to single reads; BK_SCAN_ABLATE=5 sends every N run to level2"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bronko_amd import pack_reads, Params, _ffi
_ffi.use_testing_library(True)
from tests import helpers
from oracle import oracle as orc
hp = os.path.join(ROOT, "tests", "golden", "HPV16.fa")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
ix = orc.Index.build(k, [hp])
eng = helpers.engine_from_oracle_index(ix, Params(ci=1))
reads = helpers.hpv_reads(4000, seed=20 + k)
def bad(rs):
    w, l = pack_reads(rs, k)
    eng.sample_begin(); eng.push_reads(0, w, l); res = eng.sample_finish(1)
    pile = orc.sample_pileup(ix, [rs], ci=1)
    return not all(np.array_equal(a, b) for a, b in zip(res.arrays(), pile.arrays())), res, pile
b, res, pile = bad(reads)
print("k", k, "ablate", os.environ.get("BK_SCAN_ABLATE"), "bad", b, flush=True)
if b:
    cur = reads
    n = 2
    while len(cur) > 1:
        chunk = max(1, len(cur) // n)
        reduced = False
        for i in range(0, len(cur), chunk):
            cand = cur[:i] + cur[i + chunk:]
            if cand and bad(cand)[0]:
                cur = cand; n = max(n - 1, 2); reduced = True; break
        if not reduced:
            if chunk == 1: break
            n = min(len(cur), n * 2)
    print("minimal failing set:", len(cur), "reads")
    b, res, pile = bad(cur)
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
        a, o = getattr(res, name), getattr(pile, name)
        d = np.nonzero(a != o)[0]
        if len(d):
            print("SET", name, [(int(x) // 4, "ACGT"[int(x) % 4], int(a[x]), int(o[x])) for x in d[:10]])
    print("kmer_stats", res.kmer_stats.tolist(), "oracle", pile.kmc_stats.tolist())
    os.environ["BK_L2_STATS"] = "1"
    eng2 = helpers.engine_from_oracle_index(ix, Params(ci=1))
    w2, l2 = pack_reads(cur, k)
    eng2.sample_begin(); eng2.push_reads(0, w2, l2); eng2.sample_finish(1)
    for label, rs in (("reversed", cur[::-1]), ("rotated by 1", cur[1:] + cur[:1]), ("last read first", cur[-1:] + cur[:-1]), ("plus an exact read in front", [reads[0]] + cur)):
        print(label, "bad", bad(rs)[0])
    singles = cur
    g = open(hp).read().split("\n", 1)[1].replace("\n", "")
    first_bad = None
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
        d = np.nonzero(getattr(res, name) != getattr(pile, name))[0]
        if len(d): first_bad = int(d[0]) // 4 if first_bad is None else min(first_bad, int(d[0]) // 4)
    print("first differing cell", first_bad)
    for idx, r in enumerate(singles):
        s = r.decode()
        rc = s.translate(str.maketrans("ACGT", "TGCA"))[::-1]
        best = None
        for strand, q in (("+", s), ("-", rc)):
            for a in range(0, len(q) - 24, 8):
                p = g.find(q[a:a + 24])
                if p >= 0:
                    best = (strand, p - a, q); break
            if best: break
        if best:
            strand, pos, q = best
            ref = g[pos:pos + len(q)]
            mm = [i for i in range(len(q)) if ref[i] != q[i]]
            mm_read = mm if strand == "+" else sorted(len(q) - 1 - i for i in mm)
            if pos - 5 <= first_bad <= pos + len(q) + 5:
                print("lane", idx, "read strand", strand, "ref pos", pos, "mismatches at read bases", mm_read)
        else:
            print("lane", idx, "no placement found")
