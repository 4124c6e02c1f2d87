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


def test_one_hot_bin_through_every_tier_of_the_binned_scan(oracle, hpv):
    """1.3 million copies of one read that carries one substitution, on both strands: every E item falls into two bins and every V
    item into one -- per scan workgroup ten thousand items for a bucket of 24: the bucket, the bin's extension in device memory
    (256 slots), the device-wide overflow list (a million entries) and, past its end, the plane itself (item_direct) all take
    their share of the same counters.  Depths of 900,000 and 400,000 at the substitution, every cell as the oracle says."""
    ix, eng = hpv
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    w = bytearray(g[2000:2150].upper())
    w[75] = ord("A") if w[75] != ord("A") else ord("C")
    w = bytes(w)
    reads = [w] * 900000 + [w.translate(comp)[::-1]] * 400000
    res = helpers.hip_sample(eng, [reads], 21)
    pile = oracle.sample_pileup_mt(ix, [reads], os.cpu_count() or 8)[0]
    helpers.assert_same_pileup(res, pile)
    assert int(res.fwd_depth.max()) == 900000 and int(res.rev_depth.max()) == 400000


def test_v_items_straight_into_the_regional_finalize(oracle, hpv, golden_dir, monkeypatch):
    """Round 5: a mate file whose reads are ONE scan launch never writes the V part of its plane -- the regional finalize adds the
    scan's V items up itself (FinalizeArgs::f_items) and reads from the plane only the rows Level 2 noted.  One launch (items),
    two launches of one mate file (the first launch's items are sent to the plane after all), two mate files (mate 0's are, mate 1's
    wait), a sample begun and abandoned with items waiting and rows noted, samples one after the other on one engine -- each equals
    the oracle; and the testing build with BK_NO_FUSE=1 (every launch through bin_count_kernel, as in round 4) gives the same."""
    from bronko_amd import _ffi
    ix, eng = hpv
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    a = helpers.hpv_reads(20000, seed=81, err=0.02)                  # 2 % errors: plenty of close pairs for Level 2
    b = helpers.hpv_reads(15000, seed=82, with_n=True, ragged=True)
    gm, isnv = synth.sample_genome(g, 9)
    c1, c2 = synth.paired_codes(gm, 7000, 150, 83, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    pa, pb, pab, pm = (oracle.sample_pileup(ix, [a]), oracle.sample_pileup(ix, [b]), oracle.sample_pileup(ix, [a + b]), oracle.sample_pileup(ix, mates))

    def run(e):
        helpers.assert_same_pileup(helpers.hip_sample(e, [a], 21), pa)                       # one launch
        helpers.assert_same_pileup(helpers.hip_sample(e, [b], 21), pb)                       # ... and the next sample on the same engine
        helpers.assert_same_pileup(helpers.hip_sample(e, [a + b], 21, batch=len(a)), pab)    # two launches of one mate file
        helpers.assert_same_pileup(helpers.hip_sample(e, [a + b], 21, batch=9000), pab)      # four
        helpers.assert_same_pileup(helpers.hip_sample(e, mates, 21), pm)                     # two mate files
        from bronko_amd import pack_reads
        e.sample_begin()                                                                     # abandoned with items waiting
        e.push_reads(0, *pack_reads(a, 21))
        helpers.assert_same_pileup(helpers.hip_sample(e, [b], 21), pb)
        helpers.assert_same_pileup(helpers.hip_sample(e, [a], 21), pa)

    run(eng)
    f = eng.fork()
    try:
        run(f)
    finally:
        f.close()
    monkeypatch.setenv("BK_NO_FUSE", "1")
    _ffi.use_testing_library(True)
    try:
        e2 = helpers.engine_from_oracle_index(ix)
        try:
            run(e2)
        finally:
            e2.close()
    finally:
        _ffi.use_testing_library(False)


def test_amplicon_reads_fill_every_bucket_of_a_region(oracle, sars_paths):
    """An amplicon: 600,000 reads from one 400 bp stretch of SARS-CoV-2, 2 % substitutions, both strands, pushed in three batches
    (the first launch stores its bins' V counters, the later ones add to them; the overflow list's two counters take turns).  Every
    scan workgroup puts hundreds of items into the same few bins: buckets, extensions in device memory and the overflow list all
    fill.  Against the oracle on all host cores."""
    ix = oracle.Index.build(21, sars_paths[:1])
    eng = helpers.engine_from_oracle_index(ix)
    g = synth.read_fasta_bytes(sars_paths[0])
    gm, isnv = synth.sample_genome(g[12000:12400], 9)
    codes = synth.single_end_codes(gm, 600000, 150, 77, err=0.02, isnv=isnv)
    reads = synth.BASES[codes]
    pile = oracle.sample_pileup_mt(ix, [reads], os.cpu_count() or 8)[0]
    words, lens = synth.pack_codes(codes)
    eng.sample_begin()
    for i in range(0, len(lens), 200000):
        eng.push_reads(0, words[i:i + 200000], lens[i:i + 200000])
    helpers.assert_same_pileup(eng.sample_finish(1), pile)
    eng.close()
    ix.close()


def test_lds_window_smaller_than_the_reference(oracle, golden_dir, monkeypatch, testing_lib):
    """BK_LDS_BINS (testing build) caps the LDS window: reads whose diagonal leaves it are N runs as a whole and their exact
    k-mers are counted by Level 2's membership test -- the path every genome but one of a multi-genome index takes.
    0 = no window at all: everything through Level 2."""
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    reads = helpers.hpv_reads(6000, seed=8)
    pile = oracle.sample_pileup(ix, [reads])
    for env in ({"BK_LDS_BINS": "1000"}, {"BK_LDS_BINS": "0"}, {"BK_LDS_BINS": "500"}):
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        eng = helpers.engine_from_oracle_index(ix)
        res = helpers.hip_sample(eng, [reads], 21, batch=2500)
        helpers.assert_same_pileup(res, pile)
        eng.close()
        for kk in env:
            monkeypatch.delenv(kk)
    ix.close()


def _mutated_strains(base, n, seed, n_sub):
    """n variants of `base` with n_sub seeded substitutions each (config-5 style strain set)."""
    import numpy as np
    out = []
    for s in range(n):
        g = np.frombuffer(base, np.uint8).copy()
        r = synth.splitmix64(seed * 1000 + s, 2 * n_sub)
        pos = (r[0::2] % np.uint64(len(g))).astype(np.int64)
        sh = (r[1::2] % np.uint64(3)).astype(np.int64) + 1
        for p, d in zip(pos, sh):
            g[p] = synth.BASES[(int(synth.CODE[g[p]]) + int(d)) & 3]
        out.append(("strain%02d" % s, [("seq%02d" % s, g.tobytes())]))
    return out


def test_many_strains_k31_selection_and_dirty_neighbourhoods(oracle):
    """BASELINE config 5 in miniature: 12 synthetic strains (HPV16 + ~1 % substitutions each), k = 31 (u64-wrapping
    bucket ids), reads from strain 7.  Strain-specific SNPs put most reference k-mers within Hamming distance 2 of
    another strain's k-mer, so this exercises the dirty / deferred paths of scan and finalize, and selection."""
    from bronko_amd import Params
    base = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    files = _mutated_strains(base, 12, seed=5, n_sub=80)
    ix = oracle.Index.build_mem(31, files)
    eng = helpers.engine_from_oracle_index(ix)
    gm, isnv = synth.sample_genome(files[7][1][0][1], 5)
    c1, c2 = synth.paired_codes(gm, 12000, 150, 5, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    res = helpers.hip_sample(eng, mates, 31)
    pile = oracle.sample_pileup(ix, mates)
    helpers.assert_same_pileup(res, pile)
    assert oracle.pick_best_genome(ix, res.stats.sum(axis=0), res.present.max(axis=0)) == 7
    eng.close()
    ix.close()


def test_multi_sequence_genomes_and_short_sequences(oracle):
    """Segmented genomes: several sequences per file, one shorter than k (no k-mers), diagonals must not cross
    sequence boundaries."""
    base = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    files = [("segA", [("a1", base[:3000]), ("a2", base[3000:3010]), ("a3", base[3010:6000])]),
             ("segB", [("b1", base[5000:7906]), ("b2", base[100:1500])])]
    ix = oracle.Index.build_mem(21, files)
    eng = helpers.engine_from_oracle_index(ix)
    reads = helpers.hpv_reads(15000, seed=12)
    res = helpers.hip_sample(eng, [reads], 21)
    pile = oracle.sample_pileup(ix, [reads])
    helpers.assert_same_pileup(res, pile)
    eng.close()
    ix.close()


def test_config2_full_size_one_million_reads(oracle, sars_paths):
    """BASELINE config 2 at full size: 1,000,000 x 150 bp single-end reads vs wuhan_ref k=21 (the bench workload),
    compared cell for cell with the oracle; plus the order / batching invariance the max / + votes imply."""
    ix = oracle.Index.build(21, sars_paths[:1])
    eng = helpers.engine_from_oracle_index(ix)
    ref = synth.read_fasta_bytes(sars_paths[0])
    gm, isnv = synth.sample_genome(ref, 2)
    codes = synth.single_end_codes(gm, 1000000, 150, 2 * 1000003, err=0.005, isnv=isnv)
    words, lens = synth.pack_codes(codes)
    eng.sample_begin()
    eng.push_reads(0, words, lens)
    res = eng.sample_finish(1)
    pile = oracle.sample_pileup(ix, [synth.codes_to_ascii(codes)])
    helpers.assert_same_pileup(res, pile)
    assert res.kmer_stats[0, 1] == 130000000
    # same reads, reversed and pushed in 7 uneven batches
    eng.sample_begin()
    bounds = [0, 10, 99999, 100000, 333333, 800001, 999999, 1000000]
    for a, b in zip(bounds[:-1], bounds[1:]):
        eng.push_reads(0, words[::-1][a:b], lens[::-1][a:b])
    res2 = eng.sample_finish(1)
    for x, y in zip(res.arrays(), res2.arrays()):
        assert np.array_equal(x, y)
    assert np.array_equal(res.stats, res2.stats)
    eng.close()
    ix.close()


def test_full_kmer_statistics_match_the_kmc_contract(oracle, golden_dir):
    """bk_params.full_kmer_stats: KMC's four statistics (call.rs:1190-1199) over ALL read k-mers -- including the
    ones that never touch the index (errors outside the window, flipped orientation, contamination) -- per mate."""
    from bronko_amd import Params
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    eng = helpers.engine_from_oracle_index(ix, Params(full_kmer_stats=True, kmer_table_log2=22))
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    gm, isnv = synth.sample_genome(g, 6)
    c1, c2 = synth.paired_codes(gm, 9000, 150, 6, err=0.01, isnv=isnv)
    junk = synth.codes_to_ascii(np.ascontiguousarray(synth.splitmix64(77, 500 * 150).astype(np.uint8).reshape(500, 150) & 3))
    mates = [synth.codes_to_ascii(c1) + junk + [b"ACGTNNACGT" * 12], synth.codes_to_ascii(c2)]
    res = helpers.hip_sample(eng, mates, 21)
    pile = oracle.sample_pileup(ix, mates)
    helpers.assert_same_pileup(res, pile)
    assert res.kmer_stats[:, 1].tolist() == pile.kmc_stats[:, 1].tolist()     # total k-mers
    assert res.kmer_stats[:, 2].tolist() == pile.kmc_stats[:, 2].tolist()     # unique k-mers
    assert res.kmer_stats[:, 3].tolist() == pile.kmc_stats[:, 3].tolist()     # unique counted k-mers
    assert res.kmer_stats[0, 2] > res.kmer_stats[0, 3] > 10000
    eng.close()
    # a table that starts far too small is rehashed into larger ones as the sample grows (several times here, also between the
    # batches of one mate file and on the device-packing path): same exact statistics, pileups unaffected
    for kw in ({}, {"batch": 700}, {"batch": 1500, "ascii_path": True}):
        eng = helpers.engine_from_oracle_index(ix, Params(full_kmer_stats=True, kmer_table_log2=10))
        for _ in range(2):   # (the second sample starts with the grown table)
            res = helpers.hip_sample(eng, mates, 21, **kw)
            helpers.assert_same_pileup(res, pile)
            assert res.kmer_stats[:, 2].tolist() == pile.kmc_stats[:, 2].tolist()
            assert res.kmer_stats[:, 3].tolist() == pile.kmc_stats[:, 3].tolist()
        eng.close()
    ix.close()


def test_device_side_packing_and_async_ingest(oracle, hpv):
    """bk_push_reads_ascii: K0 on the GPU (N splitting, short runs, chunking of long reads) + the 3-slot asynchronous
    ingest ring; many small batches so that slots are reused while earlier batches are still in flight."""
    ix, eng = hpv
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    long_read = g[500:7000]                                   # 6.5 kb read: chunked with k-1 overlap on the device
    reads = helpers.hpv_reads(9000, seed=14, with_n=True, ragged=True) + [b"", b"NNNN", b"ACGT", long_read,
                                                                            long_read[:3000] + b"N" + long_read[3000:]]
    pile = oracle.sample_pileup(ix, [reads])
    for batch in (None, 1000, 37):
        res = helpers.hip_sample(eng, [reads], 21, batch=batch, ascii_path=True)
        helpers.assert_same_pileup(res, pile)
        assert res.kmer_stats[0, 1] == pile.kmc_stats[0, 1]
    # host packing and device packing produce the same number of records
    host = helpers.hip_sample(eng, [reads], 21)
    assert host.kmer_stats[0, 0] == res.kmer_stats[0, 0]
    # bk_push_reads_ascii_device (ABI v7): the same lines already resident in device memory, two batches, unaligned starts
    import torch
    flat = np.frombuffer(b"".join(reads), np.uint8)
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    d_flat = torch.from_numpy(flat.copy()).to("cuda:0")
    torch.cuda.synchronize()
    eng.sample_begin()
    cut = len(reads) // 3
    for lo, hi in ((0, cut), (cut, len(reads))):
        d_off = torch.from_numpy(off[lo:hi + 1].copy()).to("cuda:0")
        torch.cuda.synchronize()
        longest = int((off[lo + 1:hi + 1] - off[lo:hi]).max())
        eng.push_reads_ascii_device(0, d_flat.data_ptr(), d_off.data_ptr(), hi - lo, int(off[hi] - off[lo]), longest)
    res2 = eng.sample_finish(1)
    helpers.assert_same_pileup(res2, pile)
    assert res2.kmer_stats[0].tolist() == res.kmer_stats[0].tolist()


def test_device_side_packing_of_reads_no_longer_than_two_words(oracle, hpv, golden_dir):
    """K0 with records of one or two words (every read of the batch <= 32 / <= 16 bases): a block of pack_words_kernel then holds
    256 reads and needs 257 offsets -- one more than it has threads (the last one was never loaded).  Several full
    blocks, reads of k .. 32 bases with N and lower case, through bk_push_reads_ascii; k = 15 with reads of 15 / 16 bases for
    the one-word record."""
    ix, eng = hpv
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    r = synth.splitmix64(4242, 3 * 3000)

    def short_reads(lo, hi, n):
        out = []
        for i in range(n):
            ln = lo + int(r[3 * i] % np.uint64(hi - lo + 1))
            at = int(r[3 * i + 1] % np.uint64(len(g) - hi)) if i % 3 else 1000 + (i % 40)   # a third pile up on one stretch (ci = 3)
            rd = bytearray(g[at:at + ln])
            if i % 97 == 0:
                rd[int(r[3 * i + 2] % np.uint64(ln))] = ord("N")
            if i % 11 == 0:
                rd = bytearray(bytes(rd).lower())
            out.append(bytes(rd))
        return out

    reads = short_reads(21, 32, 3000)
    assert max(len(x) for x in reads) == 32
    pile = oracle.sample_pileup(ix, [reads])
    assert int(pile.fwd_depth.max()) >= 3
    for batch in (None, 256, 700):
        res = helpers.hip_sample(eng, [reads], 21, batch=batch, ascii_path=True)
        helpers.assert_same_pileup(res, pile)
        assert res.kmer_stats[0].tolist() == helpers.hip_sample(eng, [reads], 21).kmer_stats[0].tolist()
    ix15 = oracle.Index.build(15, [os.path.join(golden_dir, "HPV16.fa")])
    eng15 = helpers.engine_from_oracle_index(ix15)
    try:
        reads = short_reads(15, 16, 3000)
        pile = oracle.sample_pileup(ix15, [reads])
        assert int(pile.fwd_depth.max()) >= 3
        for batch in (None, 256):
            helpers.assert_same_pileup(helpers.hip_sample(eng15, [reads], 15, batch=batch, ascii_path=True), pile)
    finally:
        eng15.close()
        ix15.close()


def test_a_fork_with_parameters_of_its_own(oracle, hpv):
    """bk_engine_fork_params (ABI v7): ci / cs / cx of a fork differ from the parent's -- each equals the oracle run with its own
    thresholds; what shapes the shared tables (n_fixed, use_full_kmer, full_kmer_stats) must equal the parent's."""
    from bronko_amd import BronkoError, Params
    ix, eng = hpv
    reads = helpers.hpv_reads(12000, seed=77, err=0.01)
    forks = [(dict(ci=1), eng.fork(Params(ci=1))), (dict(ci=5, cs=40), eng.fork(Params(ci=5, cs=40))), (dict(ci=2, cx=60), eng.fork(Params(ci=2, cx=60)))]
    try:
        for kw, f in forks:
            helpers.assert_same_pileup(helpers.hip_sample(f, [reads], 21), oracle.sample_pileup(ix, [reads], **kw))
        helpers.assert_same_pileup(helpers.hip_sample(eng, [reads], 21), oracle.sample_pileup(ix, [reads]))   # the parent keeps its own
        for bad in (Params(n_fixed=3), Params(use_full_kmer=True), Params(full_kmer_stats=True)):
            with pytest.raises(BronkoError) as ei:
                eng.fork(bad)
            assert ei.value.status == -1
    finally:
        for _, f in forks:
            f.close()


def test_reference_walk_from_global_memory(oracle, golden_dir, monkeypatch, testing_lib):
    """BK_REF_IN_LDS=0 forces the REF_LDS=false kernel (reference + flag nibbles read from global memory), the path
    indexes too large for LDS take; also combined with a tiny LDS window."""
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    reads = helpers.hpv_reads(8000, seed=21, with_n=True)
    pile = oracle.sample_pileup(ix, [reads])
    for env in ({"BK_REF_IN_LDS": "0"}, {"BK_REF_IN_LDS": "0", "BK_LDS_BINS": "100"}):
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        eng = helpers.engine_from_oracle_index(ix)
        res = helpers.hip_sample(eng, [reads], 21)
        helpers.assert_same_pileup(res, pile)
        eng.close()
        for kk in env:
            monkeypatch.delenv(kk)
    ix.close()


def test_k31_reads_that_reach_buckets_through_the_u64_wrap_of_their_ids(oracle, sars_paths):
    """k = 31 bucket ids exceed 2^64 and the reference keeps them modulo 2^64 (lcb.rs:1-45), so a read k-mer that has
    nothing to do with the reference can still land in a reference bucket: the one whose exact rank differs by 2^64.
    Build such k-mers on purpose (tests/helpers.lcb_rank / lcb_unrank), mix them with ordinary reads, and require
    the same pileup and statistics as the oracle (whose assign_buckets wraps like the reference's)."""
    ix = oracle.Index.build(31, [sars_paths[0]])
    g = synth.read_fasta_bytes(sars_paths[0])
    reads = []
    n_alias = 0
    for start in range(500, 29000, 700):
        v, _ = oracle.canonical_kmer(g[start:start + 31].decode())
        for j in range(2, 28):
            masked = v & ~(3 << (2 * (30 - j)))
            r = helpers.lcb_rank(masked, j, 31)
            other = helpers.lcb_unrank(r - (1 << 64) if r >= (1 << 64) else r + (1 << 64), 31)
            if other is None:
                continue
            av, aj = other
            n_alias += 1
            for b in ((0, 1, 2, 3) if n_alias % 5 == 0 else (n_alias & 3,)):
                x = av | (b << (2 * (30 - aj)))
                s = helpers.kmer_str(x, 31)
                if n_alias & 1:
                    s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
                reads += [s.encode()] * (3 + (n_alias % 3))
    assert n_alias > 500
    gm, isnv = synth.sample_genome(g, 9)
    reads += synth.codes_to_ascii(synth.single_end_codes(gm, 20000, 150, 19, isnv=isnv))
    pile = oracle.sample_pileup(ix, [reads])
    only_alias = oracle.sample_pileup(ix, [reads[:-20000]])
    assert int(only_alias.fwd_nk.sum() + only_alias.rev_nk.sum()) > 0   # the constructed k-mers do vote in the reference
    eng = helpers.engine_from_oracle_index(ix)
    res = helpers.hip_sample(eng, [reads], 31)
    helpers.assert_same_pileup(res, pile)
    eng.close()
    ix.close()


def test_k31_four_sarscov2_strains(oracle, sars_paths):
    """The four golden SARS-CoV-2 genomes at k = 31 (the case that first exposed the wrap-around aliasing)."""
    ix = oracle.Index.build(31, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 4)
    reads = synth.codes_to_ascii(synth.single_end_codes(gm, 60000, 150, 44, isnv=isnv))
    pile = oracle.sample_pileup(ix, [reads])
    eng = helpers.engine_from_oracle_index(ix)
    res = helpers.hip_sample(eng, [reads], 31)
    helpers.assert_same_pileup(res, pile)
    assert oracle.pick_best_genome(ix, res.stats[0], res.present[0]) == 2
    eng.close()
    ix.close()


def test_counter_planes_of_read_shards_add_up(oracle, hpv):
    """The multi-GPU identity on one device: two engines take the two halves of a sample, the
    second engine's counter planes (difference arrays, wrapping u64) are added into the first's -- what the RCCL
    all-reduce does across ranks -- and finalize of the sum equals the oracle on all reads, both mates."""
    import torch
    ix, _ = hpv

    class _Dev:   # __cuda_array_interface__ view of a device pointer (as bench.py)
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}

    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa")), 31)
    c1, c2 = synth.paired_codes(gm, 6000, 150, 31, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    pile = oracle.sample_pileup(ix, mates)
    from bronko_amd import pack_reads
    engs = [helpers.engine_from_oracle_index(ix) for _ in range(2)]
    for e in engs:
        e.sample_begin()
    for m, reads in enumerate(mates):
        half = len(reads) // 3   # uneven shards
        for e, part in zip(engs, (reads[:half], reads[half:])):
            w, l = pack_reads(part, 21)
            e.push_reads(m, w, l)
    torch.cuda.synchronize()
    for m in range(2):
        a = torch.as_tensor(_Dev(engs[0].counters_ptr(m), engs[0].counter_len), device="cuda:0")
        b = torch.as_tensor(_Dev(engs[1].counters_ptr(m), engs[1].counter_len), device="cuda:0")
        a += b
    torch.cuda.synchronize()
    res = engs[0].sample_finish(2)
    helpers.assert_same_pileup(res, pile)
    for e in engs:
        e.close()


def test_a_push_split_into_several_launches(oracle, hpv, monkeypatch, testing_lib):
    """Large pushes are cut into launches of bounded size (the 16-bit halves of the LDS difference array bound the records
    one workgroup may see); BK_MAX_LAUNCH_RECORDS forces the same splitting at test size, also for device-side packing
    where the record count is only known on the device."""
    ix, _ = hpv
    reads = helpers.hpv_reads(5000, seed=77, with_n=True)
    pile = oracle.sample_pileup(ix, [reads])
    monkeypatch.setenv("BK_MAX_LAUNCH_RECORDS", "1100")
    eng = helpers.engine_from_oracle_index(ix)
    monkeypatch.delenv("BK_MAX_LAUNCH_RECORDS")
    helpers.assert_same_pileup(helpers.hip_sample(eng, [reads], 21), pile)
    helpers.assert_same_pileup(helpers.hip_sample(eng, [reads], 21, ascii_path=True), pile)
    eng.close()


def test_reverse_complement_repeats(oracle):
    """A genome that contains the reverse complement of one of its own segments: the same canonical k-mers occur on both
    strands, so cells of the second copy do not stand in the orientation of their k-mers' first occurrence (such cells are
    not "clean": the V rows are laid out by the first occurrence) and exact hits on it land on ids seen before."""
    r = synth.splitmix64(4242, 4000)
    g = bytes(synth.BASES[int(x) & 3] for x in r)
    seg = g[600:1100]
    rc = seg[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    genome = g[:2500] + rc + g[2500:]
    files = [("inv", [("inv1", genome)])]
    ix = oracle.Index.build_mem(21, files)
    gm, isnv = synth.sample_genome(genome, 8)
    reads = synth.codes_to_ascii(synth.single_end_codes(gm, 20000, 150, 18, isnv=isnv, err=0.01))
    pile = oracle.sample_pileup(ix, [reads])
    eng = helpers.engine_from_oracle_index(ix)
    helpers.assert_same_pileup(helpers.hip_sample(eng, [reads], 21), pile)
    eng.close()
    ix.close()


@pytest.mark.parametrize("kmer_stats", [False, True])
def test_sharded_finalize_on_one_device(oracle, sars_paths, kmer_stats):
    """The cheap multi-GPU form (reduce-scatter + bk_sample_finalize_shard + max / sum of the pileups) simulated on one
    device with four engines standing for four ranks: every "rank" scans its reads; the planes are summed (what the
    reduce-scatter computes) and each rank keeps ONLY its quarter of the sum -- the rest of its plane is overwritten with
    garbage, which a correct shard never reads; the four partial pileups are combined by max / sum, the statistics by
    sum.  Result must equal the oracle on all reads (4 strains: several genomes, dirty neighbourhoods, deferred k-mers).
    kmer_stats: with bk_params.full_kmer_stats the ranks' statistics tables are exchanged first (bk_kmer_table_partition, an
    all-to-all -- here device copies --, bk_kmer_table_replace) and KMC's distinct / counted totals of the whole sample come out
    exact on every rank; without the exchange the sharded finalize refuses."""
    import torch
    from bronko_amd import pack_reads, Params
    from bronko_amd.dist import DeviceVector
    world = 4
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[1]), 13)
    c1, c2 = synth.paired_codes(gm, 20000, 150, 13, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    pile = oracle.sample_pileup(ix, mates)
    engs = [helpers.engine_from_oracle_index(ix, Params(full_kmer_stats=kmer_stats, kmer_table_log2=16)) for _ in range(world)]
    assert engs[0].counter_len % 64 == 0
    for r, e in enumerate(engs):
        e.sample_begin()
        for m, reads in enumerate(mates):
            lo, hi = len(reads) * r // world, len(reads) * (r + 1) // world
            w, l = pack_reads(reads[lo:hi], 21)
            e.push_reads(m, w, l)
    torch.cuda.synchronize()
    if kmer_stats:
        with pytest.raises(Exception, match="exchange"):
            engs[0].sample_finalize_shard(2, 0, world)
        parts = []
        for e in engs:
            kp, cp, off = e.kmer_table_partition(world)
            keys = torch.as_tensor(DeviceVector(kp, max(off[world], 1)), device="cuda:0")
            cnts = torch.as_tensor(DeviceVector(cp, max(off[world], 1), "<i4"), device="cuda:0")
            parts.append([(keys[off[r]:off[r + 1]].clone(), cnts[off[r]:off[r + 1]].clone()) for r in range(world)])
        assert sum(len(p[0]) for p in parts[0]) > 1000                 # (the case is not vacuous)
        held = []
        for r, e in enumerate(engs):                                    # "all-to-all": rank r receives group r of every rank
            rk = torch.cat([parts[q][r][0] for q in range(world)])
            rc = torch.cat([parts[q][r][1] for q in range(world)])
            torch.cuda.synchronize()
            e.kmer_table_replace(rk.data_ptr(), rc.data_ptr(), len(rk))
            held.append((rk, rc))
        torch.cuda.synchronize()
    n = engs[0].counter_len
    part = n // world
    for m in range(2):
        planes = [torch.as_tensor(DeviceVector(e.counters_ptr(m), n), device="cuda:0") for e in engs]
        total = planes[0].clone()
        for p in planes[1:]:
            total += p
        for r, p in enumerate(planes):
            p.fill_(0x5a5a5a5a5a5a5a5)
            p[r * part:(r + 1) * part] = total[r * part:(r + 1) * part]
    torch.cuda.synchronize()
    cells4 = engs[0].total_cells * 4
    piles, sums = [], []
    for r, e in enumerate(engs):
        e.sample_finalize_shard(2, r, world)
        piles.append(torch.as_tensor(DeviceVector(e.pileup_ptr(), 4 * cells4), device="cuda:0"))
        sp, sn = e.shard_sums()
        sums.append(torch.as_tensor(DeviceVector(sp, sn), device="cuda:0"))
    torch.cuda.synchronize()
    depth = torch.stack([p[:2 * cells4] for p in piles]).max(dim=0).values
    nk = torch.stack([p[2 * cells4:] for p in piles]).sum(dim=0)
    ssum = torch.stack(sums).sum(dim=0)
    piles[0][:2 * cells4] = depth
    piles[0][2 * cells4:] = nk
    sums[0].copy_(ssum)
    torch.cuda.synchronize()
    engs[0].sample_merge_shards()
    res = engs[0].sample_download(2)
    helpers.assert_same_pileup(res, pile)
    assert oracle.pick_best_genome(ix, res.stats.sum(axis=0), res.present.max(axis=0)) == 1
    assert res.kmer_stats[:, 1].tolist() == pile.kmc_stats[:, 1].tolist()
    if kmer_stats:
        assert res.kmer_stats[:, 2:4].tolist() == pile.kmc_stats[:, 2:4].tolist(), (res.kmer_stats, pile.kmc_stats)
    for e in engs:
        e.close()
    ix.close()


@pytest.mark.parametrize("width", [16, 32, 64])
def test_sharded_finalize_through_the_engine_side_transport(oracle, sars_paths, width):
    """bk_shard_transport / bk_shard_received (ABI v7): four engines stand for four ranks; every rank's plane is packed by the
    engine for the wire (16-bit lanes in int32 words, int32, or the plane itself), the reduce-scatter is simulated on the one
    device (typed wrapping sums of the four send buffers, part r to rank r), the engine widens the received part and
    bk_sample_finalize_shard maps it -- the planes themselves are never modified.  Result = the oracle's on all reads.  Then a
    sample with a k-mer count beyond the lane range of width 16 / 4 ranks: packers raise the flag, it travels with the
    statistics, bk_sample_download refuses on every rank (BK_ERR_RANGE) -- and the same sample at width 32 is the oracle's."""
    import torch
    from bronko_amd import BronkoError, pack_reads
    from bronko_amd.dist import DeviceVector, pick_width
    world = 4
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[1]), 13)
    c1, c2 = synth.paired_codes(gm, 20000, 150, 13, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    hot = bytearray(synth.read_fasta_bytes(sars_paths[1])[5000:5150])
    hot[70] = ord("A") if hot[70] != ord("A") else ord("C")               # one substitution, seen 40,000 times: a V count far above 32767 / 4
    heavy = [mates[0] + [bytes(hot)] * 40000, mates[1]]
    engs = [helpers.engine_from_oracle_index(ix) for _ in range(world)]
    def run(sample, w):
        for r, e in enumerate(engs):
            e.sample_begin()
            for m, reads in enumerate(sample):
                lo, hi = len(reads) * r // world, len(reads) * (r + 1) // world
                wd, ln = pack_reads(reads[lo:hi], 21)
                e.push_reads(m, wd, ln)
        helpers.sharded_finalize_on_one_device(engs, 2, w)

    pile = oracle.sample_pileup(ix, mates)
    for rep in range(2):                                                   # (twice: the planes are zeroed at the next sample's first push)
        run(mates, width)
        for e in engs:
            helpers.assert_same_pileup(e.sample_download(2), pile)
            assert not e.transport_overflow()
    # what bk_shard_measure says about this sample
    for e in engs:
        e.sample_begin()
        wd, ln = pack_reads(heavy[0], 21)
        e.push_reads(0, wd, ln)
    ptrs = [e.shard_measure(0) for e in engs]
    torch.cuda.synchronize()                                               # (the measuring kernels run on the engines' streams)
    mx = [torch.as_tensor(DeviceVector(p, 2), device="cuda:0").tolist() for p in ptrs]
    assert all(40000 <= m[1] <= 40010 for m in mx) and all(m[0] > 0 for m in mx), mx      # (+ the odd sequencing error that recreates the substitution)
    assert pick_width(max(m[0] for m in mx), 40000, world) == 32 and pick_width(100, 8191, 4) == 16 and pick_width(2 ** 32, 1, 1) == 64
    if width == 16:
        run(heavy, 16)
        for e in engs:
            with pytest.raises(BronkoError) as ei:
                e.sample_download(2)
            assert ei.value.status == -6                                   # BK_ERR_RANGE, on every rank
            assert not e.transport_overflow()                              # (reported once)
        run(heavy, 32)
        want = oracle.sample_pileup(ix, heavy)
        for e in engs:
            helpers.assert_same_pileup(e.sample_download(2), want)
    for e in engs:
        e.close()
    ix.close()


def test_long_reads_with_indels_and_chimeras(oracle, sars_paths):
    """Reads that leave their seed diagonal: 1000 bp reads (63-word records) with deletions, insertions and chimeric joins of
    both strands.  Level 1 proves what lies on the diagonal of the chosen seed; everything behind an indel / breakpoint is
    far from it and must come out of the general path with the same counts (the oracle does not care how reads look)."""
    ix = oracle.Index.build(21, [sars_paths[0]])
    g = synth.read_fasta_bytes(sars_paths[0])
    r = synth.splitmix64(909, 6 * 3000)
    tr = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for i in range(3000):
        a = int(r[6 * i] % np.uint64(len(g) - 1100))
        s = bytearray(g[a:a + 1000])
        kind = int(r[6 * i + 1] % np.uint64(4))
        p = 100 + int(r[6 * i + 2] % np.uint64(800))
        if kind == 0:
            del s[p:p + 1 + int(r[6 * i + 3] % np.uint64(5))]                      # deletion
        elif kind == 1:
            s[p:p] = bytes(synth.BASES[int(x) & 3] for x in r[6 * i + 3:6 * i + 5])  # 2-base insertion
        elif kind == 2:
            b = int(r[6 * i + 3] % np.uint64(len(g) - 600))
            s = s[:p] + bytearray(g[b:b + 500][::-1].translate(tr))                 # chimera: other locus, other strand
        for e in range(4):                                                           # a few substitutions
            q = int(r[6 * i + 4] >> np.uint64(8 * e)) % len(s)
            s[q] = synth.BASES[(synth.CODE[s[q]] + 1 + e % 3) & 3]
        s = bytes(s)
        reads.append(s[::-1].translate(tr) if r[6 * i + 5] & np.uint64(1) else s)
    pile = oracle.sample_pileup(ix, [reads])
    eng = helpers.engine_from_oracle_index(ix)
    helpers.assert_same_pileup(helpers.hip_sample(eng, [reads], 21), pile)
    helpers.assert_same_pileup(helpers.hip_sample(eng, [reads], 21, stride_words=20), pile)   # cut into overlapping 320-base records
    eng.close()
    ix.close()


def test_lds_window_on_any_genome_gives_the_same_counts(oracle, sars_paths, monkeypatch, testing_lib):
    """Multi-genome indexes: the LDS window (difference array + Level 1's arrays) is put on the genome the first reads vote
    for; BK_WINDOW_FILE forces it elsewhere.  Whatever genome it sits on, the counts are the same -- it is about speed."""
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[3]), 23)
    c1, c2 = synth.paired_codes(gm, 15000, 150, 23, isnv=isnv)
    mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    pile = oracle.sample_pileup(ix, mates)
    eng = helpers.engine_from_oracle_index(ix)
    helpers.assert_same_pileup(helpers.hip_sample(eng, mates, 21), pile)          # the vote (genome 3, presumably)
    for wf in ("0", "1", "2", "3"):
        monkeypatch.setenv("BK_WINDOW_FILE", wf)
        helpers.assert_same_pileup(helpers.hip_sample(eng, mates, 21), pile)
    monkeypatch.delenv("BK_WINDOW_FILE")
    assert oracle.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0)) == 3
    eng.close()
    ix.close()


def test_forked_engines_take_samples_in_turn(oracle, hpv):
    """bk_engine_fork: a second engine on the parent's device tables with its own counter planes, outputs and stream.  Three
    different samples alternate over parent and fork without any host synchronisation in between (every call is
    asynchronous on the engine's own stream: sample i+1's scan overlaps sample i's finalize); each result is downloaded at
    the end and must equal the oracle's for that sample -- twice over, so that every engine is also reused."""
    from bronko_amd import pack_reads
    ix, eng = hpv
    fork = eng.fork()
    try:
        samples = [helpers.hpv_reads(9000 + 1500 * i, seed=60 + i, err=0.004 * (i + 1)) for i in range(3)]
        packed = [pack_reads(s, 21, None) for s in samples]
        want = [oracle.sample_pileup(ix, [s]) for s in samples]
        for order in ([0, 1, 2], [2, 0, 1]):
            engines = [eng, fork]
            done = {}
            for turn, si in enumerate(order):
                e = engines[turn % 2]
                if turn >= 2:                       # the engine is about to be reused: take its previous sample's result first
                    done[order[turn - 2]] = e.sample_download(1)
                e.sample_begin()
                e.push_reads(0, *packed[si])
                e.sample_finalize(1)
            for turn in range(max(0, len(order) - 2), len(order)):
                done[order[turn]] = engines[turn % 2].sample_download(1)
            for si in order:
                helpers.assert_same_pileup(done[si], want[si])
                assert done[si].kmer_stats[0, 1] == want[si].kmc_stats[0, 1]
    finally:
        fork.close()


def test_empty_window_counts_kmers_and_touches_nothing(oracle):
    """n_fixed * 2 + 1 >= k (call.rs:1291-1300): the window slice is empty, nothing maps, KMC's k-mer total is still reported;
    a multi-file index at k = 11 (this combination once sent the window vote through tables that do not exist)."""
    from bronko_amd import Params
    base = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))[:2500]
    files = [("f%d" % i, [("s%d" % i, base[i * 300:i * 300 + 1500])]) for i in range(3)]
    ix = oracle.Index.build_mem(11, files)
    eng = helpers.engine_from_oracle_index(ix, Params(n_fixed=5, ci=1))
    reads = [base[i:i + 150] for i in range(0, 2300, 7)] + [b"ACGTACGTAC", b"", b"N" * 40]
    res = helpers.hip_sample(eng, [reads], 11)
    pile = oracle.sample_pileup(ix, [reads], n_fixed=5, ci=1)
    helpers.assert_same_pileup(res, pile)
    assert res.kmer_stats[0, 1] == pile.kmc_stats[0, 1] > 0 and int(res.fwd_nk.sum()) == 0
    eng.close()
    ix.close()


def test_randomised_indexes_and_reads():
    """tools/fuzz_parity.py: random small indexes (repeats, reverse-complement repeats, homopolymers, several sequences /
    files, k = 11..31, window variants incl. empty and full) x random read sets (20..400 bp, up to 12 % substitutions, indels,
    chimeras, foreign reads, N), ci = 1, HIP path vs oracle bit for bit.  1000 iterations here (~140 s; 60 until round 6: one seed in
    fifty found round 5's compiler-dependent fault); the kept runs of several seeds x 1000 are under profiles/rNN_fuzz.txt."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "1000", "11"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]


def _device_batches(gen, n_batches, dev):
    import torch
    out = []
    for b in range(n_batches):
        mates = gen(b)
        out.append([synth.pack_codes_torch(c) for c in mates])
    torch.cuda.synchronize()
    return out


def test_config3_full_size_ten_million_pairs(oracle, sars_paths):
    """BASELINE configs[2] at full size: the 4-strain index, 10,000,000 pairs (2 x 150 bp) derived from ON765678.1, seed 3,
    generated on the GPU in ten batches of 1 M pairs (the generator bench.py uses).  The oracle needs minutes and tens of GB
    for 20 M reads, so: (a) the first 1 M-pair batch alone against the oracle (all cores), bit for bit; (b) the whole sample
    through size-independent properties -- every k-mer scanned (2 x 10 M x 130), the result does not depend on the order in
    which the batches are pushed, depth can only grow and #k-mers only grow from (a) to the whole sample, and the selected
    genome is ON765678.1."""
    import torch
    dev = torch.device("cuda", 0)
    n, nb = 1000000, 10
    ix = oracle.Index.build(21, sars_paths)
    eng = helpers.engine_from_oracle_index(ix)
    fork = eng.fork()
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 3)
    batches = _device_batches(lambda b: synth.paired_codes_torch(gm, n, 150, 3, isnv=isnv, device=dev, row0=b * n), nb, dev)

    def run(e, order):
        e.sample_begin()
        for b in order:
            for m, (w, l) in enumerate(batches[b]):
                e.push_reads_device(m, w.data_ptr(), w.shape[1], l.data_ptr(), n)
        return e.sample_finish(2)

    # (a) one batch against the oracle
    c1, c2 = synth.paired_codes_torch(gm, n, 150, 3, isnv=isnv, device=dev, row0=0)
    mates = [synth.BASES[c.to(torch.uint8).cpu().numpy()] for c in (c1, c2)]
    pile, _ = oracle.sample_pileup_mt(ix, mates, os.cpu_count() or 8)
    first = run(eng, [0])
    helpers.assert_same_pileup(first, pile)
    # (b) the whole sample, two push orders on two engines
    full = run(eng, list(range(nb)))
    other = run(fork, list(reversed(range(nb))))
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk", "stats", "present"):
        assert np.array_equal(getattr(full, name), getattr(other, name)), name
    assert full.kmer_stats[:, 1].tolist() == [nb * n * 130, nb * n * 130]
    assert full.kmer_stats[:, 0].tolist() == [nb * n, nb * n]
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
        assert np.all(getattr(full, name) >= getattr(first, name)), name
    assert oracle.pick_best_genome(ix, full.stats.sum(axis=0), full.present.max(axis=0)) == 2
    fork.close()
    eng.close()
    ix.close()


def test_config4_two_hundred_million_reads(oracle, sars_paths):
    """BASELINE configs[3] on one GPU: wuhan_ref k = 21, ONE sample of 200,000,000 x 150 bp reads, seed 4, generated on the GPU
    in 200 batches of 1 M reads (bench.py --config 4's generator; 8.4 GB of records resident) -- 2.6 * 10^10 k-mer occurrences,
    per-k-mer counts of several 10^5, every push one launch among hundreds.  The oracle cannot follow, so:
    (a) the first batch alone against the oracle, bit for bit;
    (b) the whole sample by properties: every k-mer scanned, two push orders on parent and fork identical, nothing shrinks from
        (a) to the whole sample;
    (c) the same sample on a fork with cs = 100,000 (kmc -cs, call.rs:1173): depth = min(depth, cs) cell by cell, the cap is
        reached, #k-mers and statistics unchanged;
    (d) the sample's reads dealt to four engines standing for four ranks, sharded finalize through the engine-side transport:
        bk_shard_measure says 32 bits (a quarter of a 16-bit lane does not hold these counts), the result is the whole sample's;
        forced to 16 bits the packers raise the flag and bk_sample_download refuses on every rank."""
    import torch
    from bronko_amd import BronkoError, Params
    from bronko_amd.dist import DeviceVector, pick_width
    dev = torch.device("cuda", 0)
    n, nb = 1000000, 200
    ix = oracle.Index.build(21, [sars_paths[0]])
    eng = helpers.engine_from_oracle_index(ix)
    fork = eng.fork()
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[0]), 4)
    gen = lambda b: synth.single_end_codes_torch(gm, n, 150, 4 * 1000003 + 7919 * b, err=0.005, isnv=isnv, device=dev)   # noqa: E731
    batches = []
    for b in range(nb):
        batches.append(synth.pack_codes_torch(gen(b)))
    torch.cuda.synchronize()

    def run(e, order):
        e.sample_begin()
        for b in order:
            w, l = batches[b]
            e.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), n)
        return e.sample_finish(1)

    # (a)
    pile, _ = oracle.sample_pileup_mt(ix, [synth.BASES[gen(0).to(torch.uint8).cpu().numpy()]], os.cpu_count() or 8)
    first = run(eng, [0])
    helpers.assert_same_pileup(first, pile)
    # (b)
    full = run(eng, list(range(nb)))
    other = run(fork, list(reversed(range(nb))))
    names = ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk", "stats", "present")
    for name in names:
        assert np.array_equal(getattr(full, name), getattr(other, name)), name
    assert int(full.kmer_stats[0, 1]) == nb * n * 130 == 26000000000 and int(full.kmer_stats[0, 0]) == nb * n
    for name in names[:4]:
        assert np.all(getattr(full, name) >= getattr(first, name)), name
    top = int(max(full.fwd_depth.max(), full.rev_depth.max()))
    assert 200000 < top < 1000000                                          # counts of several 10^5, below kmc's -cs
    # (c)
    cs = 100000
    capped_eng = eng.fork(Params(cs=cs))
    capped = run(capped_eng, list(range(nb)))
    for name in ("fwd_depth", "rev_depth"):
        assert np.array_equal(getattr(capped, name), np.minimum(getattr(full, name), cs)), name
    assert int(capped.fwd_depth.max()) == cs
    for name in names[2:]:
        assert np.array_equal(getattr(capped, name), getattr(full, name)), name
    capped_eng.close()
    # (d)
    engs = [eng, fork, eng.fork(), eng.fork()]
    for r, e in enumerate(engs):
        e.sample_begin()
        for b in range(r, nb, len(engs)):
            w, l = batches[b]
            e.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), n)
    ptrs = [e.shard_measure(0) for e in engs]
    torch.cuda.synchronize()                                               # (the measuring kernels run on the engines' streams)
    mx = np.array([torch.as_tensor(DeviceVector(p, 2), device=dev).tolist() for p in ptrs]).max(axis=0)
    assert pick_width(int(mx[0]), int(mx[1]), len(engs)) == 32, mx
    helpers.sharded_finalize_on_one_device(engs, 1, 32)
    for e in engs:
        got = e.sample_download(1)
        for name in names:
            assert np.array_equal(getattr(got, name), getattr(full, name)), name
        assert int(got.kmer_stats[0, 1]) == nb * n * 130 and int(got.kmer_stats[0, 0]) == nb * n
    for r, e in enumerate(engs):
        e.sample_begin()
        for b in range(r, nb, len(engs)):
            w, l = batches[b]
            e.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), n)
    helpers.sharded_finalize_on_one_device(engs, 1, 16)
    for e in engs:
        with pytest.raises(BronkoError) as ei:
            e.sample_download(1)
        assert ei.value.status == -6                                       # BK_ERR_RANGE
    for e in reversed(engs):
        e.close()
    ix.close()


def test_config5_hundred_strains_k31(oracle, sars_paths):
    """BASELINE configs[4] shape: 100 synthetic strains (wuhan_ref + 300 substitutions each), k = 31 -- 37 M window buckets, 8 M
    alias keys of the u64 bucket-id wrap, a 2.9 GB counter plane, tables far beyond the L2 -- and 200,000 reads of one sample
    derived from strain 7, against the oracle bit for bit (pileups of all 100 genomes, per-genome statistics, selection); then the
    config's shape on forks (64 samples in turn) and one sample at its full size of 1,000,000 reads."""
    files = synth.strain_files(synth.read_fasta_bytes(sars_paths[0]), 100)
    ix = oracle.Index.build_mem(31, files)
    eng = helpers.engine_from_oracle_index(ix)
    gm, isnv = synth.sample_genome(files[7][1][0][1], 5)
    codes = synth.single_end_codes(gm, 200000, 150, 55, isnv=isnv)
    words, lens = synth.pack_codes(codes)
    eng.sample_begin()
    eng.push_reads(0, words, lens)
    res = eng.sample_finish(1)
    pile, _ = oracle.sample_pileup_mt(ix, [synth.BASES[codes]], os.cpu_count() or 8)
    helpers.assert_same_pileup(res, pile)
    assert oracle.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0)) == 7

    # what `bronko call` runs with at this size: votes for the selected genome only (two finalize passes, the sparse finalize below
    # the 2.9 GB planes), on forks of the same tables with parameters of their own (bk_engine_fork_params)
    from bronko_amd import Params
    names = ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk")

    def check_selected(got, want, best):
        assert np.array_equal(got.stats, want.stats) and np.array_equal(got.present, want.present)
        lo, n = ix.genome_cells(best)
        for name in names:
            g, w = getattr(got, name), getattr(want, name)
            assert np.array_equal(g[lo * 4:(lo + n) * 4], w[lo * 4:(lo + n) * 4]), name
            assert not g[:lo * 4].any() and not g[(lo + n) * 4:].any(), name

    sel = [eng.fork(Params(pileup_selected_only=True)) for _ in range(3)]
    sel[0].sample_begin()
    sel[0].push_reads(0, words, lens)
    check_selected(sel[0].sample_finish(1), pile, 7)
    # ... and the config's shape: 64 samples (20,000 reads each, sample s from strain s mod 100) taken in turn by the three forks,
    # nothing synchronised in between; each against the oracle
    n_s, per = 64, 20000
    want = {}

    def settle(s_j):   # the result of sample s_j, taken from its engine just before the engine is reused (or at the end)
        best = oracle.pick_best_genome(ix, want[s_j].stats.sum(axis=0), want[s_j].present.max(axis=0))
        assert best == s_j % 100, (s_j, best)
        check_selected(sel[s_j % len(sel)].sample_download(1), want.pop(s_j), best)

    for s_i in range(n_s):
        gm_s, isnv_s = synth.sample_genome(files[s_i % 100][1][0][1], 5 + s_i)
        c = synth.single_end_codes(gm_s, per, 150, 5 * 1000003 + s_i, isnv=isnv_s)
        want[s_i] = oracle.sample_pileup_mt(ix, [synth.BASES[c]], os.cpu_count() or 8)[0]
        e = sel[s_i % len(sel)]
        if s_i >= len(sel):
            settle(s_i - len(sel))
        e.sample_begin()
        e.push_reads(0, *synth.pack_codes(c))
        e.sample_finalize(1)
    for s_i in range(n_s - len(sel), n_s):
        settle(s_i)
    # ... and one sample at the config's full size: 1,000,000 reads derived from strain 31, every genome's rows on the parent
    # (call.rs:1305-1384 to the letter) and the selected genome's on a fork, both against the oracle on all host cores
    gm_f, isnv_f = synth.sample_genome(files[31][1][0][1], 36)
    c_f = synth.single_end_codes(gm_f, 1000000, 150, 5 * 1000003 + 31, isnv=isnv_f)
    w_f, l_f = synth.pack_codes(c_f)
    pile_f, _ = oracle.sample_pileup_mt(ix, [synth.BASES[c_f]], os.cpu_count() or 8)
    assert oracle.pick_best_genome(ix, pile_f.stats.sum(axis=0), pile_f.present.max(axis=0)) == 31
    eng.sample_begin()
    eng.push_reads(0, w_f, l_f)
    helpers.assert_same_pileup(eng.sample_finish(1), pile_f)
    sel[1].sample_begin()
    sel[1].push_reads(0, w_f, l_f)
    check_selected(sel[1].sample_finish(1), pile_f, 31)
    for e in sel:
        e.close()
    eng.close()
    ix.close()


def test_votes_for_the_selected_genome_only(oracle, sars_paths):
    """bk_params.pileup_selected_only (what `bronko call` runs with): two finalize passes -- statistics of every genome, the genome
    picked on the device, then votes for that genome alone.  Statistics, presence flags and the selected genome's rows equal the
    oracle's; every other genome's rows stay zero.  Four SARS-CoV-2 strains (k = 21, paired) and twelve HPV16 strains (k = 31)."""
    from bronko_amd import Params
    cases = []
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[1]), 71)
    c1, c2 = synth.paired_codes(gm, 20000, 150, 71, isnv=isnv)
    cases.append((ix, [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)], 21, 1))
    files = _mutated_strains(synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa")), 12, 60, 90)
    ix2 = oracle.Index.build_mem(31, files)
    gm, isnv = synth.sample_genome(files[5][1][0][1], 72)
    cases.append((ix2, [synth.codes_to_ascii(synth.single_end_codes(gm, 15000, 150, 72, isnv=isnv))], 31, 5))
    for ix, mates, k, want in cases:
        pile = oracle.sample_pileup(ix, mates)
        best = oracle.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0))
        assert best == want
        eng = helpers.engine_from_oracle_index(ix, Params(pileup_selected_only=True))
        for rep in range(2):
            res = helpers.hip_sample(eng, mates, k)
            assert np.array_equal(res.stats, pile.stats) and np.array_equal(res.present, pile.present)
            lo, n = ix.genome_cells(best)
            for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
                got, ref = getattr(res, name), getattr(pile, name)
                assert np.array_equal(got[lo * 4:(lo + n) * 4], ref[lo * 4:(lo + n) * 4]), name
                assert not got[:lo * 4].any() and not got[(lo + n) * 4:].any(), name
        # ... and the device's calls on those rows are the oracle's
        recs, out, nrec, nmaj, nmin, breadth, depth = oracle.call_variants(ix, best, pile, oracle.default_call_params(k))
        eng.sample_call(len(mates))
        summ, drecs = eng.download_calls()
        assert (summ.file_id, summ.n_records, summ.n_major, summ.n_minor) == (best, nrec, nmaj, nmin)
        oracle.lib().orc_free(out)
        eng.close()
        ix.close()


@pytest.mark.parametrize("selected_only", [False, True])
@pytest.mark.parametrize("n_strains", [100, 140])
def test_file_bitmaps_and_the_paths_without_them(oracle, monkeypatch, testing_lib, selected_only, n_strains):
    """With up to 128 genome files the engine describes a bucket by the set of files in it (IndexView::ent_files) and takes the
    statistics, the selected genome's votes and -- on an index with touch lists (forced here) -- every genome's votes of the
    reference k-mers from those bitmaps and the tables built on them (own occurrences cell by cell, the rest from per-k-mer
    lists); with more files, or a bucket that holds a file twice (the repeat planted here), it walks the entries as before.
    100 and 140 strains of a 1.2 kb fragment of HPV16 with a direct repeat, k = 21: both equal the oracle, statistics and rows."""
    from bronko_amd import Params
    monkeypatch.setenv("BK_SPARSE_FINALIZE", "1")
    base = bytearray(synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))[1000:2200])
    base[700:760] = base[100:160]                                # every strain holds these k-mers twice
    files = _mutated_strains(bytes(base), n_strains, 61, 12)
    ix = oracle.Index.build_mem(21, files)
    gm, isnv = synth.sample_genome(files[37][1][0][1], 73)
    mates = [synth.codes_to_ascii(synth.single_end_codes(gm, 6000, 150, 73, isnv=isnv))]
    pile = oracle.sample_pileup(ix, mates)
    best = oracle.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0))
    eng = helpers.engine_from_oracle_index(ix, Params(pileup_selected_only=selected_only))
    for rep in range(2):
        res = helpers.hip_sample(eng, mates, 21)
        assert np.array_equal(res.stats, pile.stats) and np.array_equal(res.present, pile.present)
        if not selected_only:
            helpers.assert_same_pileup(res, pile)
            continue
        lo, n = ix.genome_cells(best)
        for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
            got, ref = getattr(res, name), getattr(pile, name)
            assert np.array_equal(got[lo * 4:(lo + n) * 4], ref[lo * 4:(lo + n) * 4]), name
            assert not got[:lo * 4].any() and not got[(lo + n) * 4:].any(), name
    eng.close()
    ix.close()


@pytest.mark.parametrize("force_sparse", [True, False])
@pytest.mark.parametrize("selected_only", [False, True])
def test_planes_are_clean_between_samples(oracle, sars_paths, monkeypatch, testing_lib, selected_only, force_sparse):
    """No counter plane is zeroed wholesale between samples: finalize clears what it read -- K2a the V counters as it reads them
    (dense planes), or the touch lists' rows (an index above 16 M counters, bk_engine.cpp alloc_sample_state; BK_SPARSE_FINALIZE
    forces that path here).  Four SARS-CoV-2 strains with paired reads and twelve HPV16 strains at k = 31 (pseudo k-mers); the
    engine is used for a sample that is begun, pushed and abandoned, then for two whole samples: each equals the oracle's."""
    from bronko_amd import Params, pack_reads
    if force_sparse:
        monkeypatch.setenv("BK_SPARSE_FINALIZE", "1")
    cases = []
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 81)
    c1, c2 = synth.paired_codes(gm, 15000, 150, 81, isnv=isnv)
    cases.append((ix, [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)], 21))
    files = _mutated_strains(synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa")), 12, 60, 91)
    ix2 = oracle.Index.build_mem(31, files)
    gm, isnv = synth.sample_genome(files[7][1][0][1], 82)
    cases.append((ix2, [synth.codes_to_ascii(synth.single_end_codes(gm, 12000, 150, 82, isnv=isnv))], 31))
    for ix, mates, k in cases:
        pile = oracle.sample_pileup(ix, mates)
        best = oracle.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0))
        eng = helpers.engine_from_oracle_index(ix, Params(pileup_selected_only=selected_only))
        eng.sample_begin()                                   # abandoned: its counts must not leak into the next sample
        words, lens = pack_reads(mates[0][:3000], k, None)
        eng.push_reads(0, words, lens)
        for rep in range(2):
            res = helpers.hip_sample(eng, mates, k)
            assert np.array_equal(res.stats, pile.stats) and np.array_equal(res.present, pile.present)
            if not selected_only:
                helpers.assert_same_pileup(res, pile)
                continue
            lo, n = ix.genome_cells(best)
            for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
                got, ref = getattr(res, name), getattr(pile, name)
                assert np.array_equal(got[lo * 4:(lo + n) * 4], ref[lo * 4:(lo + n) * 4]), name
                assert not got[:lo * 4].any() and not got[(lo + n) * 4:].any(), name
        if force_sparse:
            with pytest.raises(Exception):
                eng.counters_ptr(0)                          # sharding one sample's reads needs the dense plane
        eng.close()
        ix.close()


@pytest.mark.parametrize("paired", [False, True])
def test_every_genomes_rows_by_table_and_cell_by_cell(oracle, monkeypatch, testing_lib, paired):
    """Every genome's rows of a many-genome index (touch lists forced) are cast three ways: through the table of voters per
    (reference k-mer, window position) (voter_table_kernel + gather_table_kernel, with its bitmap of touched V rows), cell by cell
    (gather_votes_kernel<false>: BK_NO_VOTE_TABLE), and cell by cell with 64-bit maxima (-cs above 2^32: gather_votes_kernel<true>,
    which -cs >= 2^28 selects).  30 strains of HPV16 at k = 31 (alias keys, merged buckets) with one or two mate files: each equals
    the oracle (call.rs:1305-1418), cell for cell, and the statistics."""
    from bronko_amd import Params
    monkeypatch.setenv("BK_SPARSE_FINALIZE", "1")
    base = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))[:4000]
    files = _mutated_strains(base, 30, 17, 40)
    ix = oracle.Index.build_mem(31, files)
    gm, isnv = synth.sample_genome(files[11][1][0][1], 19)
    if paired:
        c1, c2 = synth.paired_codes(gm, 9000, 150, 19, isnv=isnv)
        mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
    else:
        mates = [synth.codes_to_ascii(synth.single_end_codes(gm, 15000, 150, 19, isnv=isnv))]
    results = []
    for env, cs in (({}, None), ({"BK_NO_VOTE_TABLE": "1"}, None), ({}, 1 << 33)):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        prm = Params() if cs is None else Params(cs=cs)
        pile = oracle.sample_pileup(ix, mates) if cs is None else oracle.sample_pileup(ix, mates, cs=cs)
        eng = helpers.engine_from_oracle_index(ix, prm)
        for rep in range(2):                                 # (twice: the table and the bitmap are per sample)
            res = helpers.hip_sample(eng, mates, 31)
            helpers.assert_same_pileup(res, pile)
        results.append(res)
        eng.close()
        for k_ in env:
            monkeypatch.delenv(k_)
    for x, y in zip(results[0].arrays(), results[1].arrays()):
        assert np.array_equal(x, y)
    ix.close()


@pytest.mark.parametrize("shape", ["one genome", "every genome's rows", "selected genome"])
def test_no_device_allocation_after_an_engines_first_sample(oracle, monkeypatch, testing_lib, shape):
    """Everything an engine needs for a sample is allocated with it (bk_engine_create / bk_engine_fork) or, for the buffers sized
    by a launch's records, by its first sample: from the second sample on bk_device_memory reports the same free bytes after every
    sample, on the engine and on a fork -- no hipMalloc inside a sample (it synchronises the device: siblings in flight stall, and
    round 5's table of voters, allocated inside the first every-genome finalize of every fork, sat in bench.py's timed region).
    The three shapes: one genome (binned scan, regional finalize), a many-genome index with touch lists and gathered votes for every
    genome's rows (the table of voters) and for the selected genome only."""
    from bronko_amd import Params
    from bronko_amd.engine import device_memory
    if shape == "one genome":
        ix = oracle.Index.load(os.path.join(helpers.GOLDEN, "hpv.bkdb"))
        g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
        k, prm = 21, Params()
    else:
        monkeypatch.setenv("BK_SPARSE_FINALIZE", "1")
        base = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))[:4000]
        files = _mutated_strains(base, 12, 17, 40)
        ix = oracle.Index.build_mem(31, files)
        g = files[3][1][0][1]
        k, prm = 31, Params(pileup_selected_only=(shape == "selected genome"))
    gm, isnv = synth.sample_genome(g, 23)
    mates = [synth.codes_to_ascii(synth.single_end_codes(gm, 6000, 150, 23, isnv=isnv))]
    pile = oracle.sample_pileup(ix, mates)
    eng = helpers.engine_from_oracle_index(ix, prm)
    fork = eng.fork()
    for e in (eng, fork):
        res = helpers.hip_sample(e, mates, k)            # the first sample: record-sized buffers
        if shape != "selected genome":
            helpers.assert_same_pileup(res, pile)
    free0 = device_memory(0)[0]
    for rep in range(3):
        for e in (eng, fork):
            helpers.hip_sample(e, mates, k)
            assert device_memory(0)[0] == free0, (shape, rep)
    fork.close()
    eng.close()
    ix.close()


@pytest.mark.parametrize("frac", [0.9, 0.5])
def test_samples_that_are_mostly_not_the_references_reads(oracle, hpv, frac):
    """Most of a real viral sample is the host's reads.  Level 2 takes reads marked whole (no diagonal) 64 at a time, rolls them and
    strikes every k-mer neither of whose halves is a reference k-mer's half before it is looked up (level2_kernel take_records, the
    pigeonhole filters HalfView::bits).  20,000 reads of which 90 % / 50 % are random -- a quarter of those with 25 bases of the
    reference spliced in, whose k-mers and their neighbours DO touch the index --, the rest HPV16's with errors: every cell and
    statistic equals the oracle's (call.rs:1257-1434), through the product build."""
    ix, eng = hpv
    g = synth.read_fasta_bytes(os.path.join(helpers.GOLDEN, "HPV16.fa"))
    rng = np.random.default_rng(606 + int(frac * 10))
    gm, isnv = synth.sample_genome(g, 31)
    n = 20000
    codes = synth.single_end_codes(gm, n, 150, 31, err=0.01, isnv=isnv)
    foreign = rng.random(n) < frac
    rnd = rng.integers(0, 4, size=(n, 150), dtype=np.uint8)
    gc = synth.CODE[np.frombuffer(bytes(g), np.uint8)]
    for i in np.nonzero(foreign)[0]:
        codes[i] = rnd[i]
        if rng.random() < 0.25:
            q, a0 = int(rng.integers(0, 125)), int(rng.integers(0, len(gc) - 25))
            codes[i, q:q + 25] = gc[a0:a0 + 25]
    mates = [synth.codes_to_ascii(codes)]
    pile = oracle.sample_pileup(ix, mates)
    for rep in range(2):
        res = helpers.hip_sample(eng, mates, 21)
        helpers.assert_same_pileup(res, pile)


def test_two_hundred_and_fifty_strains(oracle, sars_paths):
    """"Hundreds of strains against hundreds of samples" (/root/reference/README.md:12; the loop call.rs:212-294 over build.rs:145-231's
    index): 250 synthetic strains at k = 31 -- 79 M window slots, a 5.5 GB counter plane, more genome files than a file bitmap holds
    (128: the statistics pass walks the BucketInfo lists) -- and 20,000 reads of a sample derived from strain 189, every genome's
    rows (gathered cell by cell: bk_gather.hip needs no bitmap) and then the selected genome's on a fork, against the oracle bit for
    bit."""
    from bronko_amd import Params
    files = synth.strain_files(synth.read_fasta_bytes(sars_paths[0]), 250)
    ix = oracle.Index.build_mem(31, files)
    eng = helpers.engine_from_oracle_index(ix)
    gm, isnv = synth.sample_genome(files[189][1][0][1], 9)
    codes = synth.single_end_codes(gm, 20000, 150, 77, isnv=isnv)
    words, lens = synth.pack_codes(codes)
    pile, _ = oracle.sample_pileup_mt(ix, [synth.BASES[codes]], os.cpu_count() or 8)
    for rep in range(2):                                   # (twice: what a sample touched is clean again)
        eng.sample_begin()
        eng.push_reads(0, words, lens)
        res = eng.sample_finish(1)
        helpers.assert_same_pileup(res, pile)
    best = oracle.pick_best_genome(ix, pile.stats.sum(axis=0), pile.present.max(axis=0))
    assert best == 189
    sel = eng.fork(Params(pileup_selected_only=True))
    sel.sample_begin()
    sel.push_reads(0, words, lens)
    got = sel.sample_finish(1)
    assert np.array_equal(got.stats, pile.stats) and np.array_equal(got.present, pile.present)
    lo, n = ix.genome_cells(best)
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
        g, w = getattr(got, name), getattr(pile, name)
        assert np.array_equal(g[lo * 4:(lo + n) * 4], w[lo * 4:(lo + n) * 4]), name
        assert not g[:lo * 4].any() and not g[(lo + n) * 4:].any(), name
    sel.close()
    eng.close()
    ix.close()


def test_level2_on_as_many_workgroups_as_its_marks_are_worth(oracle, sars_paths):
    """With four engines in a family a single-genome engine's Level 2 works on marked records / 256 workgroups, at least half the
    CU count (l2_plan_kernel, ScanArgs::l2_plan): the same pileups as the oracle's and as the engine's before it had siblings,
    for reads that mark one record in a hundred, for reads that mark nearly all of them (5 % errors, foreign reads) and for none."""
    ref = sars_paths[0]
    ix = oracle.Index.build(21, [ref])
    eng = helpers.engine_from_oracle_index(ix)
    g = synth.read_fasta_bytes(ref)
    gm, isnv = synth.sample_genome(g, 5)
    rng = np.random.default_rng(11)
    sets = {
        "on target": synth.codes_to_ascii(synth.single_end_codes(gm, 60000, 150, 21, err=0.005, isnv=isnv)),
        "5 % errors": synth.codes_to_ascii(synth.single_end_codes(gm, 30000, 150, 22, err=0.05, isnv=isnv)),
        "half foreign": synth.codes_to_ascii(synth.single_end_codes(gm, 15000, 150, 23, err=0.01, isnv=isnv))
                        + [bytes(b"ACGT"[i] for i in rng.integers(0, 4, 150)) for _ in range(15000)],
        "error free": synth.codes_to_ascii(synth.single_end_codes(gm, 20000, 150, 24, err=0.0, isnv=isnv)),
    }
    alone = {name: helpers.hip_sample(eng, [reads], 21) for name, reads in sets.items()}
    for name, reads in sets.items():
        helpers.assert_same_pileup(alone[name], oracle.sample_pileup(ix, [reads]))
    forks = [eng.fork(), eng.fork(), eng.fork()]                          # a family of four: the plan is on
    for e in (eng, forks[1]):
        for name, reads in sets.items():
            got = helpers.hip_sample(e, [reads], 21)
            for x, y in zip(got.arrays(), alone[name].arrays()):
                assert np.array_equal(x, y), name
            assert np.array_equal(got.stats, alone[name].stats) and np.array_equal(got.kmer_stats, alone[name].kmer_stats), name
    for e in forks:
        e.close()
    eng.close()
    ix.close()
