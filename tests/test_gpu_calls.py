"""Reference selection + baseline noise + variant calling on the device (bk_sample_call) against the oracle's
host restatement of call.rs:422-502 / :799-967 / :969-1150: the same genome, the same records -- integers and AF bit for bit
(one correctly rounded division), SOR to the last few ulps (the device's ln) -- and the same coverage summary."""
import os

import numpy as np
import pytest

from bronko_amd import synth
from tests import helpers

pytestmark = pytest.mark.gpu


def _compare(oracle, ix, eng, mates, k, **overrides):
    res = helpers.hip_sample(eng, mates, k)
    pile = oracle.sample_pileup(ix, mates)
    helpers.assert_same_pileup(res, pile)
    best = oracle.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0))
    op = oracle.default_call_params(k)
    dp = eng.call_params()
    for name, v in overrides.items():
        setattr(op, name, v)
        setattr(dp, name, v)
    recs, out, n, nmaj, nmin, breadth, depth = oracle.call_variants(ix, best, pile, op)
    eng.sample_call(len(mates), dp)
    summ, drecs = eng.download_calls()
    assert summ.file_id == best
    assert (summ.n_records, summ.n_major, summ.n_minor) == (n, nmaj, nmin)
    assert summ.covered / summ.positions == breadth
    assert (summ.coverage / summ.covered if summ.covered else float("nan")) == depth or (summ.covered == 0 and np.isnan(depth))
    assert len(drecs) == n
    for d, o in zip(drecs, recs):
        assert (d.seq_id, d.pos, d.ref_base, d.alt_base) == (o["seq_id"], o["pos"], o["ref_base"], o["alt_base"])
        assert (d.fwd_ref, d.rev_ref, d.fwd_alt, d.rev_alt, d.depth) == (o["fwd_ref"], o["rev_ref"], o["fwd_alt"], o["rev_alt"], o["depth"])
        assert d.af == o["af"]
        assert abs(d.sor - o["sor"]) <= 1e-12 * max(1.0, abs(o["sor"]))
        assert "%.3f" % d.sor == "%.3f" % o["sor"]
    oracle.lib().orc_free(out)
    # Noise.max of every position of the selected genome, bit for bit (every sequence of the genome is its own walk)
    if best >= 0:
        lo, ncell = ix.genome_cells(best)
        want = []
        for s_lo, s_n in ix.sequence_cells(best):
            want.append(oracle.baseline_noise(pile.fwd_depth[s_lo * 4:(s_lo + s_n) * 4], pile.rev_depth[s_lo * 4:(s_lo + s_n) * 4])[0])
        want = np.concatenate(want) if want else np.zeros(0)
        got = eng.download_noise()
        assert got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64)), \
            "noise differs at %s" % np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0][:5]
    return n


def test_hpv_single_end_calls(oracle, golden_dir):
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    eng = helpers.engine_from_oracle_index(ix)
    reads = helpers.hpv_reads(60000, seed=3)
    assert _compare(oracle, ix, eng, [reads], 21) > 0
    # the filters' other branches
    _compare(oracle, ix, eng, [reads], 21, no_end_filter=1, no_strand_balance_filter=1)
    _compare(oracle, ix, eng, [reads], 21, no_strand_filter=1, min_depth=10, min_af=0.01)
    eng.close()
    ix.close()


def test_four_strains_paired_selection_and_calls(oracle, sars_paths):
    ix = oracle.Index.build(21, sars_paths)
    eng = helpers.engine_from_oracle_index(ix)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 31)
    c1, c2 = synth.paired_codes(gm, 120000, 150, 31, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    assert _compare(oracle, ix, eng, mates, 21) > 10
    eng.close()
    ix.close()


def test_noise_walk_taken_apart_and_in_one_wave(oracle, golden_dir, sars_paths, monkeypatch, testing_lib):
    """get_baseline_noise (call.rs:799-967) with its two chains in two waves (sums; table with the pass-by tests) + the strip per
    position, and the walk in one wave as rounds 2-4 ran it (BK_NOISE_SERIAL, testing build): Noise.max bit for bit either way
    (_compare), for HPV16 deep and shallow (stretches without coverage, few values per window, many equal frequencies) and a
    SARS-CoV-2 sample."""
    cases = []
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    cases.append((ix, [helpers.hpv_reads(60000, seed=3)]))
    cases.append((ix, [helpers.hpv_reads(1500, seed=4, err=0.03)]))
    cases.append((ix, [helpers.hpv_reads(300, seed=5, err=0.05)]))
    ix2 = oracle.Index.build(21, [sars_paths[0]])
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[0]), 41)
    cases.append((ix2, [synth.codes_to_ascii(synth.single_end_codes(gm, 200000, 150, 41, isnv=isnv))]))
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("BK_NOISE_SERIAL", "1")
        for cix, mates in cases:
            eng = helpers.engine_from_oracle_index(cix)
            _compare(oracle, cix, eng, mates, 21, min_depth=10, min_af=0.01)
            eng.close()
    ix.close()
    ix2.close()


def test_no_reads_no_genome(oracle, golden_dir):
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    eng = helpers.engine_from_oracle_index(ix)
    eng.sample_begin()
    eng.sample_finalize(1)
    eng.sample_call(1)
    summ, recs = eng.download_calls()
    assert summ.file_id == -1 and summ.n_records == 0 and recs == []
    eng.close()
    ix.close()


def test_calls_need_a_finalized_sample_with_the_same_mates(oracle, golden_dir):
    """bk_sample_call refuses (BK_ERR_STATE) on an engine that never finalized a sample -- its pileup is uninitialised memory -- and
    with another number of mate files than the finalize had; a fork starts as empty as a new engine."""
    from bronko_amd import BronkoError
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    eng = helpers.engine_from_oracle_index(ix)
    fork = eng.fork()
    for e in (eng, fork):
        with pytest.raises(BronkoError) as ei:
            e.sample_call(1)
        assert ei.value.status == -5
    reads = helpers.hpv_reads(3000, seed=5)
    helpers.hip_sample(eng, [reads], 21)
    with pytest.raises(BronkoError) as ei:
        eng.sample_call(2)
    assert ei.value.status == -5
    eng.sample_call(1)
    summ, recs = eng.download_calls()
    assert summ.file_id == 0 and summ.n_records == len(recs)
    eng.sample_begin()                       # a sample that was begun: nothing to call yet
    with pytest.raises(BronkoError):
        eng.sample_call(1)
    fork.close()
    eng.close()
    ix.close()
