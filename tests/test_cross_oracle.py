"""The C oracle (oracle/bronko_oracle.c) against a second restatement written separately from the Rust text
(oracle/cross_oracle.py: plain Python, k-mers as strings, dictionaries, 64-bit masking spelled out): the fixtures under
tests/golden/call_*.npz were produced by that script in the build container; here the C oracle must reproduce every cell,
statistic and KMC figure.  Cases: the derived HPV16 SNP known answer of SURVEY.md §8c, seeded HPV16 reads (errors, N, lower
case, short reads), a paired 4-strain sample, and k = 31 (bucket ids wrap modulo 2^64) with n_fixed = 3."""
import glob
import os

import numpy as np
import pytest

from tests import helpers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "call_*.npz")) if not p.endswith("call_fuzz.npz"))


def _expand(z, name):
    a = np.zeros(int(z["n_cells4"]), np.uint64)
    a[z[name + "_idx"]] = z[name + "_val"]
    return a


def test_fixtures_are_present():
    assert len(CASES) == 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-4] for p in CASES])
def test_c_oracle_reproduces_the_python_restatement(oracle, path):
    z = np.load(path)
    genomes = [os.path.join(GOLDEN, str(g)) for g in z["genomes"]]
    k, n_fixed, ci, n_mates = int(z["k"]), int(z["n_fixed"]), int(z["ci"]), int(z["n_mates"])
    mates = [bytes(z["reads%d" % m]).split(b"\n") for m in range(n_mates)]
    ix = oracle.Index.build(k, genomes)
    pile = oracle.sample_pileup(ix, mates, n_fixed=n_fixed, ci=ci)
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
        want = _expand(z, name)
        got = getattr(pile, name)
        assert np.array_equal(got, want), (name, int((got != want).sum()))
    assert np.array_equal(pile.stats, z["stats"])
    assert np.array_equal(pile.present, z["present"])
    assert np.array_equal(pile.kmc_stats, z["kmc"])
    # the multi-threaded orchestration bench.py times gives the same
    got, _ = oracle.sample_pileup_mt(ix, mates, 4, n_fixed=n_fixed, ci=ci)
    assert np.array_equal(got.fwd_depth, _expand(z, "fwd_depth")) and np.array_equal(got.stats, z["stats"])
    ix.close()


def test_known_answer_of_the_hpv_snp():
    """SNP A->T at 0-based 1000; the mutant base carries depth 10 / 7 with 6 + 6 k-mers, the mirrored
    reference-base votes of the canonical == true branch appear as singletons; perfect 0, variant 30 of 42 k-mers."""
    z = np.load(os.path.join(GOLDEN, "call_hpv_snp.npz"))
    fd, rd, fk, rk = (_expand(z, n) for n in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"))
    cell = 1000 * 4 + 3
    assert (fd[cell], rd[cell], fk[cell], rk[cell]) == (10, 7, 6, 6)
    rows = sorted(set(int(i) // 4 for i in np.nonzero(fd)[0]))
    assert rows == [986, 988, 990, 992, 996, 998, 1000, 1008, 1010, 1012]
    assert z["stats"].tolist() == [[[0, 30, 0]]]
    assert z["kmc"].tolist() == [[357, 357, 42, 42]]


def test_c_oracle_reproduces_the_fuzz_fixture(oracle):
    """tests/golden/call_fuzz.npz (oracle/cross_oracle.py --fuzz): 200 small cases in the shape of tools/fuzz_parity.py's --
    repeats, reverse-complement repeats, low complexity, 1-8 files of 1-3 sequences, k 11-31, n_fixed 0 / 1 / 2 / 5 incl. the empty
    window, --use-full-kmer, ci 1-3, indels, chimeras, foreign reads, N, lower case, one or two mate files -- computed by the
    second restatement; the C oracle must reproduce every cell, statistic and KMC figure of every one of them."""
    z = np.load(os.path.join(GOLDEN, "call_fuzz.npz"))
    n = int(z["n_cases"])
    assert n >= 200
    for c in range(n):
        files, k, mates, kw, pre = helpers.fuzz_fixture_case(z, c)
        # (sequence names: the first whitespace token of the id line, build.rs:178-182)
        ix = oracle.Index.build_mem(k, [(fn, [(rid.decode().split()[0], sq) for rid, sq in seqs]) for fn, seqs in files])
        pile = oracle.sample_pileup(ix, mates, **kw)
        for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
            assert np.array_equal(getattr(pile, name), helpers.expand_sparse(z, name, pre)), (c, name)
        assert np.array_equal(pile.stats, z[pre + "stats"]), c
        assert np.array_equal(pile.present, z[pre + "present"]), c
        assert np.array_equal(pile.kmc_stats, z[pre + "kmc"]), c
        ix.close()
