"""map_kmers restatement (call.rs:1257-1434): derived known answer + structural properties.

The reference has no test or golden output for `call`, so these are *derived* checks:
the first one is the hand-analysed case recorded in SURVEY.md §8(c); the others are properties that follow
from the cited lines.
"""
import os

import numpy as np

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def revcomp(s):
    return s.translate(COMP)[::-1]


def _hpv(oracle, golden_dir):
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    seq = ix.files()[0][1][0][1]
    return ix, seq


def test_derived_snp_known_answer(oracle, golden_dir):
    """HPV16 k=21 n_fixed=2, SNP A->T at 0-based 1000; 21 mutant windows x10, their RCs x7."""
    ix, seq = _hpv(oracle, golden_dir)
    assert seq[1000:1001] == b"A"
    mut = seq[:1000] + b"T" + seq[1001:]
    fw = [mut[s:s + 21] for s in range(980, 1001)]
    kmers = [oracle.kmer_to_u64(x) for x in fw] + [oracle.kmer_to_u64(revcomp(x)) for x in fw]
    counts = [10] * 21 + [7] * 21
    pile = oracle.Pileup(ix)
    oracle.map_kmers(ix, kmers, counts, pile)
    fd = pile.fwd_depth.reshape(-1, 4)
    rd = pile.rev_depth.reshape(-1, 4)
    fk = pile.fwd_nk.reshape(-1, 4)
    rk = pile.rev_nk.reshape(-1, 4)
    want = {986: 1, 988: 2, 990: 2, 992: 3, 996: 2, 998: 3, 1000: 3, 1008: 2, 1010: 0, 1012: 0}  # pos -> base
    nz = sorted(set(np.nonzero(fd.sum(1) + rd.sum(1))[0].tolist()))
    assert nz == sorted(want)
    for pos, b in want.items():
        exp_f = [0, 0, 0, 0]
        exp_f[b] = 10
        exp_r = [0, 0, 0, 0]
        exp_r[b] = 7
        assert fd[pos].tolist() == exp_f and rd[pos].tolist() == exp_r
        nk = 6 if pos == 1000 else 1
        assert fk[pos][b] == nk and rk[pos][b] == nk and fk[pos].sum() == nk
    assert pile.stats[0, 0].tolist() == [0, 30, 0]     # perfect 0, variant 30 of 42, unique 0
    assert pile.present[0, 0] == 1
    ix.close()


def test_perfect_kmers_vote_reference_base(oracle, golden_dir):
    """Every reference k-mer with count n (both strands) => depth n at each window position with the ref base."""
    ix, seq = _hpv(oracle, golden_dir)
    k = 21
    starts = range(100, 400)
    kmers = [oracle.kmer_to_u64(seq[s:s + k]) for s in starts] + [oracle.kmer_to_u64(revcomp(seq[s:s + k])) for s in starts]
    counts = [5] * len(starts) + [9] * len(starts)
    pile = oracle.Pileup(ix)
    oracle.map_kmers(ix, kmers, counts, pile)
    fd = pile.fwd_depth.reshape(-1, 4)
    rd = pile.rev_depth.reshape(-1, 4)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    for pos in range(130, 380):
        b = code[seq[pos]]
        assert fd[pos][b] == 5 and rd[pos][b] == 9, pos
        assert fd[pos].sum() == 5 and rd[pos].sum() == 9
    assert pile.stats[0, 0].tolist() == [600, 0, 600]
    ix.close()


def test_window_slice_and_use_full_kmer(oracle, golden_dir):
    ix, seq = _hpv(oracle, golden_dir)
    k = 21
    km = [oracle.kmer_to_u64(seq[2000:2000 + k])]
    fwd_is_canon = not oracle.canonical_kmer(seq[2000:2000 + k])[1]
    for n_fixed, full, lo, hi in ((2, False, 2, 18), (0, False, 0, 20), (5, False, 5, 15), (2, True, 0, 21), (10, False, 0, 0)):
        pile = oracle.Pileup(ix)
        oracle.map_kmers(ix, km, [4], pile, n_fixed=n_fixed, use_full_kmer=full)
        tot = (pile.fwd_depth + pile.rev_depth).reshape(-1, 4).sum(1)
        nz = np.nonzero(tot)[0]
        if hi == lo:
            assert len(nz) == 0
            continue
        # the window is applied in canonical orientation: mirror it when the ref k-mer was rc'ed
        if fwd_is_canon:
            exp = list(range(2000 + lo, 2000 + hi))
        else:
            exp = list(range(2000 + lo, 2000 + hi))  # vote lands at location + idx in both branches (call.rs:1334,1361)
        assert nz.tolist() == exp
    ix.close()


def test_depth_is_max_and_nk_adds_across_mates(oracle, golden_dir):
    """call.rs:316-317: R1 then R2 mapped into the same arrays: depth = max, #kmers adds."""
    ix, seq = _hpv(oracle, golden_dir)
    k = 21
    km = [oracle.kmer_to_u64(seq[3000:3000 + k])]
    pile = oracle.Pileup(ix, n_mates=2)
    oracle.map_kmers(ix, km, [4], pile, mate=0)
    oracle.map_kmers(ix, km, [9], pile, mate=1)
    d = (pile.fwd_depth + pile.rev_depth).reshape(-1, 4)
    n = (pile.fwd_nk + pile.rev_nk).reshape(-1, 4)
    assert d[3010].max() == 9 and n[3010].max() == 2
    assert pile.stats[:, 0, 0].tolist() == [1, 1]
    ix.close()


def test_kmc_contract_counting(oracle):
    reads = [b"ACGTACGTAC", b"ACGTNACGTA", b"acgtacg", b"AC"]
    km, ct, st = oracle.count_kmers(5, reads, ci=1)
    d = {int(a): int(b) for a, b in zip(km, ct)}
    u = oracle.kmer_to_u64
    # read1: ACGTA CGTAC GTACG TACGT ACGTA CGTAC ; read2: (ACGT|N|ACGTA) -> ACGTA ; read3: ACGTA CGTAC GTACG
    assert d == {u("ACGTA"): 4, u("CGTAC"): 3, u("GTACG"): 2, u("TACGT"): 1}
    assert st == [4, 10, 4, 4]
    km, ct, st = oracle.count_kmers(5, reads, ci=3, cs=3)
    d = {int(a): int(b) for a, b in zip(km, ct)}
    assert d == {u("ACGTA"): 3, u("CGTAC"): 3} and st[3] == 2      # -ci3 drops, -cs3 saturates
    km, ct, st = oracle.count_kmers(5, reads, ci=1, cx=3)
    assert u("ACGTA") not in set(int(x) for x in km)               # -cx excludes on the true count


def test_multithreaded_orchestration_equals_the_single_threaded_one(oracle, sars_paths):
    """bronko_oracle_mt.c (bench.py's cpu_baseline: sharded counting + map_kmers over chunks in parallel, call.rs:1279-1281)
    gives the literal single-threaded result bit for bit, single-end and paired, any thread count."""
    from bronko_amd import synth
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 9)
    c1, c2 = synth.paired_codes(gm, 6000, 150, 9, isnv=isnv)
    c1[5, 40] = 0
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    mates[0][7] = mates[0][7][:60] + b"N" + mates[0][7][61:]     # a read split by N
    mates[1][3] = b"ACGT"                                          # shorter than k
    ref = oracle.sample_pileup(ix, mates, ci=2)
    for threads in (1, 3, 8):
        got, secs = oracle.sample_pileup_mt(ix, mates, threads, ci=2)
        for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk", "stats", "present", "kmc_stats"):
            assert np.array_equal(getattr(ref, name), getattr(got, name)), (threads, name)
        assert secs[0] >= 0 and secs[1] >= 0
    # 2-D symbol arrays are accepted as well (what bench.py hands over)
    got, _ = oracle.sample_pileup_mt(ix, [synth.BASES[c1], synth.BASES[c2]], 4, ci=2)
    ref2 = oracle.sample_pileup(ix, [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)], ci=2)
    assert np.array_equal(ref2.fwd_depth, got.fwd_depth) and np.array_equal(ref2.stats, got.stats)
    ix.close()
