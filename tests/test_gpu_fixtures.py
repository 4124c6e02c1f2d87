"""The HIP path, through the C ABI, DIRECTLY on the committed fixtures of the second restatement (oracle/cross_oracle.py, plain
Python from the Rust text; tests/golden/call_*.npz): reads -> K0 -> scan -> finalize -> four arrays, statistics, presence, KMC
totals == the fixture, with no C oracle in between.  tests/test_cross_oracle.py holds the C oracle to the same files on the CPU, so
the engine and its everyday checker are each pinned to a reading of call.rs:1257-1434 / lcb.rs:1-45 that is not the other's."""
import glob
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
GOLDEN = helpers.GOLDEN
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "call_*.npz")) if not p.endswith("call_fuzz.npz"))


def _same(res, z, pre, what, full_stats):
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
        got, want = getattr(res, name), helpers.expand_sparse(z, name, pre)
        assert np.array_equal(got, want), (what, name, int((got != want).sum()))
    assert np.array_equal(res.stats, z[pre + "stats"]), (what, res.stats.tolist(), z[pre + "stats"].tolist())
    assert np.array_equal(res.present, z[pre + "present"]), what
    kmc = z[pre + "kmc"]
    # KMC's figures (call.rs:1190-1199): total reads = sequences that hold a k-mer run; total k-mers; with the statistics table
    # also the distinct and the counted k-mers
    assert res.kmer_stats[:, 1].tolist() == kmc[:, 1].tolist(), (what, "total k-mers")
    if full_stats:
        assert res.kmer_stats[:, 2:4].tolist() == kmc[:, 2:4].tolist(), (what, "distinct / counted k-mers", res.kmer_stats.tolist(), kmc.tolist())


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-4] for p in CASES])
def test_hip_on_the_committed_fixtures(path):
    from bronko_amd import Params
    from bronko_amd.hostlib import HostIndex
    z = np.load(path)
    genomes = [os.path.join(GOLDEN, str(g)) for g in z["genomes"]]
    k, n_fixed, ci, n_mates = int(z["k"]), int(z["n_fixed"]), int(z["ci"]), int(z["n_mates"])
    mates = [bytes(z["reads%d" % m]).split(b"\n") for m in range(n_mates)]
    ix = HostIndex.build(k, genomes)
    for full_stats, ascii_path in ((False, False), (True, True)):
        eng = ix.engine(Params(ci=ci, n_fixed=n_fixed, full_kmer_stats=full_stats, kmer_table_log2=20))
        res = helpers.hip_sample(eng, mates, k, ascii_path=ascii_path)
        _same(res, z, "", (os.path.basename(path), full_stats), full_stats)
        eng.close()
    ix.close()


def test_hip_on_the_fuzz_fixture():
    """tests/golden/call_fuzz.npz: 200 small cases where the fuzzer goes (repeats, reverse-complement repeats, 1-8 files, k 11-31,
    n_fixed 0 / 1 / 2 / 5, --use-full-kmer, ci 1-3, N, indels, one or two mate files), each through the engine the way a host
    drives it -- packed records or sequence lines packed on the device, pushes in one piece or in batches, with and without the
    k-mer statistics table."""
    from bronko_amd import Params
    from bronko_amd.hostlib import HostIndex
    z = np.load(os.path.join(GOLDEN, "call_fuzz.npz"))
    n = int(z["n_cases"])
    assert n >= 200
    for c in range(n):
        files, k, mates, kw, pre = helpers.fuzz_fixture_case(z, c)
        ix = HostIndex.build_mem(k, [(fn, [(rid.decode().split()[0], sq) for rid, sq in seqs]) for fn, seqs in files])
        full_stats = c % 3 == 0
        eng = ix.engine(Params(ci=kw["ci"], n_fixed=kw["n_fixed"], use_full_kmer=kw["use_full_kmer"], full_kmer_stats=full_stats, kmer_table_log2=16))
        res = helpers.hip_sample(eng, mates, k, batch=(17 if c % 4 == 1 else None), ascii_path=(c % 2 == 1))
        _same(res, z, pre, ("case %d" % c, k, kw), full_stats)
        eng.close()
        ix.close()
