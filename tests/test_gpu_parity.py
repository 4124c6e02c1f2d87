"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.
Bit-exact (integer work): four pileup arrays, per-genome (perfect, variant, unique), presence flags."""
import os

import numpy as np
import pytest

from bronko_amd import synth
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hpv(oracle, golden_dir):
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    eng = helpers.engine_from_oracle_index(ix)
    yield ix, eng
    eng.close()
    ix.close()


def test_hpv_single_end_matches_oracle(oracle, hpv):
    ix, eng = hpv
    reads = helpers.hpv_reads(20000, seed=1)
    res = helpers.hip_sample(eng, [reads], 21)
    pile = oracle.sample_pileup(ix, [reads])
    helpers.assert_same_pileup(res, pile)
    assert int(res.fwd_depth.max()) > 50                      # the case is not vacuous
    assert res.stats[0, 0, 0] > 10000 and res.stats[0, 0, 1] > 100
    assert res.kmer_stats[0, 1] == pile.kmc_stats[0, 1]       # k-mer occurrences scanned


def test_hpv_ragged_reads_with_n(oracle, hpv):
    """Reads shorter than k, empty reads, N inside reads (KMC splits the read there)."""
    ix, eng = hpv
    reads = helpers.hpv_reads(8000, seed=2, with_n=True, ragged=True) + [b"", b"N" * 30, b"ACGT", b"acgtn" * 40]
    res = helpers.hip_sample(eng, [reads], 21)
    pile = oracle.sample_pileup(ix, [reads])
    helpers.assert_same_pileup(res, pile)
    assert res.kmer_stats[0, 1] == pile.kmc_stats[0, 1]


def test_hpv_paired_end_shared_arrays(oracle, hpv):
    """call.rs:301-317: per-mate thresholds, R1 then R2 mapped into the same arrays."""
    ix, eng = hpv
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    gm, isnv = synth.sample_genome(g, 5)
    c1, c2 = synth.paired_codes(gm, 6000, 150, 11, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    res = helpers.hip_sample(eng, mates, 21)
    pile = oracle.sample_pileup(ix, mates)
    helpers.assert_same_pileup(res, pile)
    assert res.stats.shape == (2, 1, 3) and res.stats[1, 0, 0] > 0


def test_thresholds_ci_cs_cx(oracle, golden_dir):
    """-ci drops, -cs saturates the reported count, -cx drops on the true count."""
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    reads = helpers.hpv_reads(5000, seed=3, err=0.01)
    for ci, cs, cx in ((1, 1000000, 1000000000), (5, 20, 1000000000), (2, 1000000, 40)):
        from bronko_amd import Params
        eng = helpers.engine_from_oracle_index(ix, Params(ci=ci, cs=cs, cx=cx))
        res = helpers.hip_sample(eng, [reads], 21)
        pile = oracle.sample_pileup(ix, [reads], ci=ci, cs=cs, cx=cx)
        helpers.assert_same_pileup(res, pile)
        if cs == 20:
            assert int(res.fwd_depth.max()) == 20
        eng.close()
    ix.close()


def test_window_params(oracle, golden_dir):
    """--n-fixed / --use-full-kmer window slices (call.rs:1291-1300), including the empty window."""
    from bronko_amd import Params
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    reads = helpers.hpv_reads(3000, seed=4)
    for n_fixed, full in ((0, False), (5, False), (2, True), (10, False)):
        eng = helpers.engine_from_oracle_index(ix, Params(n_fixed=n_fixed, use_full_kmer=full))
        res = helpers.hip_sample(eng, [reads], 21)
        pile = oracle.sample_pileup(ix, [reads], n_fixed=n_fixed, use_full_kmer=full)
        helpers.assert_same_pileup(res, pile)
        eng.close()
    ix.close()


def test_batched_pushes_equal_one_push(oracle, hpv):
    ix, eng = hpv
    reads = helpers.hpv_reads(5000, seed=6)
    a = helpers.hip_sample(eng, [reads], 21)
    b = helpers.hip_sample(eng, [reads], 21, batch=777)
    for x, y in zip(a.arrays(), b.arrays()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.stats, b.stats)


def test_sarscov2_four_strains_selection(oracle, sars_paths):
    """Config 3 shape: 4-strain index, sample derived from ON765678.1 => selection must pick file 2."""
    ix = oracle.Index.build(21, sars_paths)
    eng = helpers.engine_from_oracle_index(ix)
    g = synth.read_fasta_bytes(sars_paths[2])
    gm, isnv = synth.sample_genome(g, 3)
    c1, c2 = synth.paired_codes(gm, 15000, 150, 3, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    res = helpers.hip_sample(eng, mates, 21)
    pile = oracle.sample_pileup(ix, mates)
    helpers.assert_same_pileup(res, pile)
    tot = res.stats.sum(axis=0)
    assert oracle.pick_best_genome(ix, tot, res.present.max(axis=0)) == 2
    eng.close()
    ix.close()


def test_k_variants(oracle, golden_dir):
    """k = 15 (MIN), 19 (tests/build_tests.rs:23-35) and 31 (MAX, u64-wrapping bucket ids)."""
    hp = os.path.join(golden_dir, "HPV16.fa")
    for k in (15, 19, 31):
        ix = oracle.Index.build(k, [hp])
        eng = helpers.engine_from_oracle_index(ix)
        reads = helpers.hpv_reads(4000, seed=20 + k)
        res = helpers.hip_sample(eng, [reads], k)
        pile = oracle.sample_pileup(ix, [reads])
        helpers.assert_same_pileup(res, pile)
        eng.close()
        ix.close()


def test_degenerate_identical_reads_spill_the_lds_histogram(oracle, hpv):
    """Millions of occurrences of the same reference k-mers: the packed 16-bit LDS bins must spill, not wrap.
    300k copies of one 150 bp reference window (+ its reverse complement) and an all-A homopolymer read set."""
    ix, eng = hpv
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    w = g[1000:1150]
    reads = [w] * 200000 + [w.translate(comp)[::-1]] * 100000 + [b"A" * 150] * 1000
    res = helpers.hip_sample(eng, [reads], 21)
    pile = oracle.sample_pileup(ix, [reads])
    helpers.assert_same_pileup(res, pile)
    assert int(res.fwd_depth.max()) == 200000 and int(res.rev_depth.max()) == 100000


def test_overflow_planes_when_lds_histogram_is_smaller_than_the_reference(oracle, golden_dir, monkeypatch):
    """BK_LDS_BINS caps the LDS histogram so that most reference k-mers take the XCD-private plane path (the
    path large multi-genome indexes use); BK_NO_XCD_PLANES=1 then forces plain agent-scope atomics."""
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    reads = helpers.hpv_reads(6000, seed=8)
    pile = oracle.sample_pileup(ix, [reads])
    for env in ({"BK_LDS_BINS": "1000"}, {"BK_LDS_BINS": "0"}, {"BK_LDS_BINS": "500", "BK_NO_XCD_PLANES": "1"}):
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        eng = helpers.engine_from_oracle_index(ix)
        res = helpers.hip_sample(eng, [reads], 21, batch=2500)
        helpers.assert_same_pileup(res, pile)
        eng.close()
        for kk in env:
            monkeypatch.delenv(kk)
    ix.close()
