"""Pin the CPU oracle against every golden artefact the reference holds for this path:
  * the two assign_buckets known answers (/root/reference/src/lcb.rs:146-154),
  * test_data/hpv.bkdb  <=>  build_indexes(test_data/HPV16.fa, k=21)  (complete index incl. in-bucket order),
  * the three `bronko build` invocations of tests/build_tests.rs (exit-code-only upstream; here: they build).
"""
import os

import numpy as np


def test_assign_buckets_astring(oracle):  # lcb.rs:146-149
    assert oracle.assign_buckets(0, 4) == [1, 2, 3, 4]


def test_assign_buckets_kstring(oracle):  # lcb.rs:151-154
    want = [238258108556, 47877379752, 215381104296, 227729135272, 235782198952, 237342480040, 238258108557,
            238236915369, 238248449705, 238254544553, 238258108558, 238257944234, 238258089642, 238258095018,
            238258106282, 238258108559, 238258108483, 238258108525, 238258108547]
    assert oracle.assign_buckets(41547505179, 19) == want


def test_assign_buckets_is_a_bijection_small_k(oracle):
    # ids are a collision-free rank of (wildcard position, other k-1 bases) in [1, k*4^(k-1)]
    for k in (3, 4, 5, 6):
        seen = {}
        for kmer in range(4 ** k):
            ids = oracle.assign_buckets(kmer, k)
            for j, b in enumerate(ids):
                masked = kmer & ~(3 << (2 * (k - 1 - j)))
                assert 1 <= b <= k * 4 ** (k - 1)
                assert seen.setdefault(b, (j, masked)) == (j, masked)
        assert len(seen) == k * 4 ** (k - 1)


def test_lcb_primitives(oracle):
    assert oracle.kmer_to_u64("ACGT") == 0b00011011          # lcb.rs:67-74 MSB first
    assert oracle.kmer_to_u64("acgtN") == 0b0001101100       # lower case ok, non-ACGT -> 0 (lcb.rs:53)
    assert oracle.reverse_complement_u64(oracle.kmer_to_u64("AAC"), 3) == oracle.kmer_to_u64("GTT")
    v, rc = oracle.canonical_kmer("GTT")                      # lcb.rs:87-95
    assert (v, rc) == (oracle.kmer_to_u64("AAC"), True)
    v, rc = oracle.canonical_kmer("AAC")
    assert (v, rc) == (oracle.kmer_to_u64("AAC"), False)


def test_hpv_bkdb_decodes_to_exact_eof_and_known_shape(oracle, golden_dir):
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    assert ix.k == 21 and ix.meta_k == 21
    assert ix.n_buckets == 165603
    assert ix.n_entries == (7906 - 21 + 1) * 21   # 165,606
    files = ix.files()
    assert len(files) == 1 and files[0][0] == "HPV16"
    assert [(n, len(s)) for n, s in files[0][1]] == [("HPV16REF", 7906)]
    off = ix.bucket_off()
    sizes = np.diff(off)
    assert (sizes == 2).sum() == 3 and (sizes == 1).sum() == 165600
    ent = ix.entries()
    two = np.nonzero(sizes == 2)[0]
    locs = sorted(tuple(int(x) for x in ent["location"][off[b]:off[b + 1]]) for b in two)
    assert locs == [(4184, 4188), (4185, 4189), (4186, 4190)]
    ix.close()


def test_build_hpv16_reproduces_hpv_bkdb(oracle, golden_dir):
    """decode(hpv.bkdb) == build_indexes(HPV16.fa, 21): ids, offsets, every BucketInfo, in-bucket order, metadata."""
    gold = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    mine = oracle.Index.build(21, [os.path.join(golden_dir, "HPV16.fa")])
    assert np.array_equal(gold.bucket_ids(), mine.bucket_ids())
    assert np.array_equal(gold.bucket_off(), mine.bucket_off())
    assert gold.entries().tobytes() == mine.entries().tobytes()
    assert gold.files() == mine.files()
    gold.close()
    mine.close()


def test_bkdb_roundtrip(oracle, golden_dir, tmp_path):
    gold = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    p = str(tmp_path / "rt.bkdb")
    gold.save(p)
    assert os.path.getsize(p) == os.path.getsize(os.path.join(golden_dir, "hpv.bkdb"))  # same varint widths
    back = oracle.Index.load(p)
    assert np.array_equal(gold.bucket_ids(), back.bucket_ids())
    assert gold.entries().tobytes() == back.entries().tobytes()
    assert gold.files() == back.files() and back.k == 21 and back.meta_k == 21
    gold.close()
    back.close()


def test_build_tests_rs_invocations(oracle, golden_dir, sars_paths):
    """tests/build_tests.rs:8-47 -- (a) 4 SARS-CoV-2 FASTAs k=21, (b) HPV16 k=19, (c) HPV16 default k."""
    a = oracle.Index.build(21, sars_paths)
    assert [f[0] for f in a.files()] == ["wuhan_ref", "OM223929.1", "ON765678.1", "PX392231.1"]
    assert [len(f[1][0][1]) for f in a.files()] == [29903, 29767, 29818, 29694]
    assert a.n_entries == 2501142 and a.n_buckets == 703025
    assert a.total_cells == 119182
    b = oracle.Index.build(19, [os.path.join(golden_dir, "HPV16.fa")])
    assert b.k == 19 and b.n_entries == (7906 - 19 + 1) * 19
    ent = b.entries()
    off = b.bucket_off()
    # every entry of a bucket shares idx and idx < k
    assert ent["idx"].max() == 18
    first_idx = ent["idx"][off[:-1]]
    assert np.array_equal(np.repeat(first_idx, np.diff(off).astype(np.int64)), ent["idx"])
    a.close()
    b.close()


def test_clean_sample_id(oracle):  # util.rs:30-50
    assert oracle.clean_sample_id("/x/y/rep1_R1.fastq.gz") == "rep1_R1"
    assert oracle.clean_sample_id("a.fq") == "a"
    assert oracle.clean_sample_id("a.fq.fq") == "a"          # trim_end_matches strips repeatedly
    assert oracle.clean_sample_id("a.fa.gz") == "a.fa"       # ".fa.gz" is not in the suffix list -> file_stem
    assert oracle.clean_sample_id("sample.txt") == "sample"


def test_exact_rank_matches_assign_buckets(oracle):
    """tests/helpers.lcb_rank (exact integers) reduces to the oracle's assign_buckets modulo 2^64 -- exhaustively for
    k = 5, and for seeded k = 31 k-mers, where ranks exceed 2^64 and wrap; lcb_unrank inverts it."""
    from bronko_amd import synth
    from tests import helpers
    for v in range(4 ** 5):
        ids = oracle.assign_buckets(v, 5)
        for j in range(5):
            masked = v & ~(3 << (2 * (4 - j)))
            r = helpers.lcb_rank(masked, j, 5)
            assert r == int(ids[j])
            assert helpers.lcb_unrank(r, 5) == (masked, j)
    wrapped = 0
    for v in synth.splitmix64(77, 300):
        v = int(v) & ((1 << 62) - 1)
        ids = oracle.assign_buckets(v, 31)
        for j in range(31):
            masked = v & ~(3 << (2 * (30 - j)))
            r = helpers.lcb_rank(masked, j, 31)
            assert r % (1 << 64) == int(ids[j])
            assert helpers.lcb_unrank(r, 31) == (masked, j)
            wrapped += r >= (1 << 64)
    assert wrapped > 0
