"""The C++ caller stages (product: selection, noise filter, variant calls, VCF / pileup-TSV writers) against the
oracle's restatement of call.rs:422-502, :792-1150, :648-695, :735-774 -- byte-identical files, identical numbers.
Inputs are oracle-made pileups of synthetic samples (the GPU is not needed for these stages)."""
import os

import numpy as np
import pytest

from bronko_amd import hostlib, synth
from bronko_amd.hostlib import HostIndex
from tests import helpers


@pytest.fixture(scope="module")
def hpv_case(oracle, golden_dir):
    p = os.path.join(golden_dir, "hpv.bkdb")
    oix = oracle.Index.load(p)
    hix = HostIndex.load(p)
    reads = helpers.hpv_reads(40000, seed=1, err=0.005)
    pile = oracle.sample_pileup(oix, [reads])
    return oix, hix, pile


def test_noise_filter_matches_oracle(oracle, hpv_case):
    oix, hix, pile = hpv_case
    want, _, _ = oracle.baseline_noise(pile.fwd_depth, pile.rev_depth)
    got = hostlib.baseline_noise_max(pile.fwd_depth, pile.rev_depth)
    assert np.array_equal(want, got)
    assert (want > 0).sum() > 1000


@pytest.mark.parametrize("variant", ["default", "no_end", "no_strand", "no_balance", "loose"])
def test_calls_and_files_match_oracle(oracle, hpv_case, tmp_path, variant):
    oix, hix, pile = hpv_case
    op = oracle.default_call_params(21)
    hp = hostlib.default_call_params(21)
    for prm in (op, hp):
        if variant == "no_end":
            prm.no_end_filter = 1
        elif variant == "no_strand":
            prm.no_strand_filter = 1
        elif variant == "no_balance":
            prm.no_strand_balance_filter = 1
            prm.strand_balance_ratio = 0.3
        elif variant == "loose":
            prm.min_af = 0.005
            prm.min_depth = 10
            prm.min_variant_depth = 1
            prm.n_per_strand = 1
            prm.variant_multiplier = 1.0
            prm.strand_odds_max = 8.0
    recs, ptr, n, nmaj, nmin, br, dc = oracle.call_variants(oix, 0, pile, op)
    o_vcf, o_tsv = str(tmp_path / "o.vcf"), str(tmp_path / "o.tsv")
    oracle.write_vcf(o_vcf, "dir/sample_R1.fastq.gz", oix, 0, ptr, n)
    oracle.write_pileup(o_tsv, oix, 0, pile)
    h_vcf, h_tsv = str(tmp_path / "h.vcf"), str(tmp_path / "h.tsv")
    hn, hmaj, hmin, hbr, hdc = hostlib.call_and_write(hix, 0, pile.arrays(), hp, h_vcf, "dir/sample_R1.fastq.gz", h_tsv)
    assert (hn, hmaj, hmin) == (n, nmaj, nmin)
    assert hbr == br and hdc == dc
    assert open(h_vcf, "rb").read() == open(o_vcf, "rb").read()
    assert open(h_tsv, "rb").read() == open(o_tsv, "rb").read()
    if variant in ("default", "loose"):
        assert nmaj >= 15          # the 20 fixed SNPs of the synthetic sample (minus the end-filtered ones)
    if variant == "loose":
        assert nmin > 5


def test_selection_matches_oracle_on_four_strains(oracle, sars_paths):
    oix = oracle.Index.build(21, sars_paths)
    hix = HostIndex.build(21, sars_paths)
    g = synth.read_fasta_bytes(sars_paths[3])
    gm, isnv = synth.sample_genome(g, 9)
    reads = synth.codes_to_ascii(synth.single_end_codes(gm, 20000, 150, 9, isnv=isnv))
    pile = oracle.sample_pileup(oix, [reads])
    want = oracle.pick_best_genome(oix, pile.stats[0], pile.present[0])
    assert want == 3
    assert hostlib.pick_best_genome(hix, pile.stats[0], pile.present[0]) == want
    # nothing present -> None (call.rs:230-233 exits 1)
    assert hostlib.pick_best_genome(hix, np.zeros((4, 3), np.uint64), np.zeros(4, np.uint8)) == -1
    # tie -> lowest file id (documented deterministic choice)
    st = np.zeros((4, 3), np.uint64)
    st[1, 0] = st[2, 0] = 100
    assert hostlib.pick_best_genome(hix, st, np.ones(4, np.uint8)) == oracle.pick_best_genome(oix, st, np.ones(4, np.uint8))


def test_clean_sample_id_matches_oracle(oracle):
    for p in ("/x/y/rep1_R1.fastq.gz", "a.fq", "a.fq.fq", "a.fa.gz", "sample.txt", "b.fasta", "c.fnq.gz", "noext"):
        assert hostlib.clean_sample_id(p) == oracle.clean_sample_id(p)
