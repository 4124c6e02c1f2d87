"""End-to-end drop-in check (BASELINE config 1 shape): the `bronko` binary on real .fastq.gz files against the
oracle's orchestration of call.rs:212-387 -- VCF and pileup TSV must be byte-identical."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from bronko_amd import synth
from tests import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BRONKO = os.path.join(ROOT, "bronko_amd", "bin", "bronko")


def write_fastq_gz(path, reads, tag):
    with gzip.open(path, "wb", compresslevel=1) as f:
        for i, r in enumerate(reads):
            f.write(b"@%s_%d\n%s\n+\n%s\n" % (tag.encode(), i, r, b"I" * len(r)))


def oracle_outputs(oracle, ix, mates, out_dir, reads_path, k=21, **kw):
    pile = oracle.sample_pileup(ix, mates, **kw)
    stats = pile.stats.sum(axis=0)
    best = oracle.pick_best_genome(ix, stats, pile.present.max(axis=0))
    recs, ptr, n, nmaj, nmin, br, dc = oracle.call_variants(ix, best, pile, oracle.default_call_params(k))
    stem = oracle.clean_sample_id(reads_path)
    oracle.write_vcf(os.path.join(out_dir, stem + ".vcf"), reads_path, ix, best, ptr, n)
    oracle.write_pileup(os.path.join(out_dir, stem + ".tsv"), ix, best, pile)
    kept = int(pile.kmc_stats[:, 3].sum())
    unmapped = kept - int(stats[best, 0]) - int(stats[best, 1])     # call.rs:242 / :336
    return stem, best, n, (int(stats[best, 0]), int(stats[best, 1]), unmapped, nmaj, nmin, br, dc)


def test_call_paired_hpv_fastq_gz(oracle, golden_dir, tmp_path):
    """2 x 20,000 synthetic HPV16 pairs (+N symbols in some reads) as .fastq.gz, prebuilt golden hpv.bkdb."""
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    gm, isnv = synth.sample_genome(g, 1, n_snp=5, n_isnv=5)
    c1, c2 = synth.paired_codes(gm, 20000, 150, 1, isnv=isnv)
    r1, r2 = synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)
    r1[7] = r1[7][:40] + b"N" + r1[7][41:]
    r2[9] = b"n" * 150
    p1, p2 = str(tmp_path / "rep1_R1.fastq.gz"), str(tmp_path / "rep1_R2.fastq.gz")
    write_fastq_gz(p1, r1, "r1")
    write_fastq_gz(p2, r2, "r2")
    out = str(tmp_path / "out")
    res = subprocess.run([BRONKO, "call", "-d", os.path.join(golden_dir, "hpv.bkdb"), "-1", p1, "-2", p2, "--pileup",
                          "-o", out, "-t", "2"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    odir = str(tmp_path / "oracle")
    os.makedirs(odir)
    stem, best, n, ov_want = oracle_outputs(oracle, ix, [r1, r2], odir, p1)
    assert stem == "rep1_R1" and n >= 5
    for ext in (".vcf", ".tsv"):
        assert open(os.path.join(out, stem + ext), "rb").read() == open(os.path.join(odir, stem + ext), "rb").read(), ext
    ov = open(os.path.join(out, "bronko_overview.tsv")).read().splitlines()
    assert ov[0].split("\t")[:2] == ["filename", "selected_genome"] and ov[1].split("\t")[:2] == [p1, "HPV16"]
    # bronko_overview.tsv (call.rs:698-732) incl. num_unmapped_kmers, which needs KMC's "unique counted k-mers"
    perfect, variant, unmapped, nmaj, nmin, br, dc = ov_want
    assert ov[1].split("\t")[2:] == [str(nmaj), str(nmin), "%.4f" % br, "%.4f" % dc, str(perfect), str(variant), str(unmapped)]
    ix.close()


def test_call_single_end_with_genomes_flag(oracle, sars_paths, tmp_path):
    """`-g` builds the index on the fly (call.rs:170-178); 4 strains, sample derived from strain 1 (OM223929.1)."""
    g = synth.read_fasta_bytes(sars_paths[1])
    gm, isnv = synth.sample_genome(g, 4)
    reads = synth.codes_to_ascii(synth.single_end_codes(gm, 30000, 150, 4, isnv=isnv))
    fq = str(tmp_path / "s1.fq")
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@s_%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    out = str(tmp_path / "o")
    res = subprocess.run([BRONKO, "call", "-g"] + sars_paths + ["-r", fq, "--pileup", "-o", out, "-t", "2"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    ix = oracle.Index.build(21, sars_paths)
    odir = str(tmp_path / "oracle")
    os.makedirs(odir)
    stem, best, n, _ = oracle_outputs(oracle, ix, [reads], odir, fq)
    assert best == 1 and stem == "s1"
    for ext in (".vcf", ".tsv"):
        assert open(os.path.join(out, stem + ext), "rb").read() == open(os.path.join(odir, stem + ext), "rb").read(), ext
    ix.close()


def test_cli_error_conventions(golden_dir, tmp_path):
    """exit code 1 + a log line for the reference's error cases (call.rs:46-134, :193-197)."""
    db = os.path.join(golden_dir, "hpv.bkdb")
    fq = str(tmp_path / "x.fastq")
    open(fq, "w").write("@a\nACGT\n+\nIIII\n")
    cases = [
        ["call", "-d", db, "-r", fq, "-k", "20"],                       # even k
        ["call", "-d", db, "-r", str(tmp_path / "x.txt")],              # not a fastq suffix
        ["call", "-r", fq],                                             # neither -d nor -g
        ["call", "-d", db, "-g", os.path.join(golden_dir, "HPV16.fa"), "-r", fq],   # both
        ["call", "-d", db, "-r", fq, "-k", "19"],                       # db k mismatch
        ["call", "-d", db, "-1", fq],                                   # unpaired pairs
        ["call", "-d", db, "-r", fq, "--min-af", "1.5"],
        ["build", "-g", str(tmp_path / "x.txt")],                       # not a fasta suffix
    ]
    for c in cases:
        res = subprocess.run([BRONKO] + c + ["-o", str(tmp_path / "e")], capture_output=True, text=True)
        assert res.returncode == 1, (c, res.stdout, res.stderr)
        assert "ERROR" in res.stdout


def test_call_alignment_mfa(oracle, golden_dir, tmp_path):
    """--alignment (call.rs:504-628): four HPV16 samples with different fixed SNPs (one of them covering too little of the
    genome: breadth < 0.90, left out) -> OUT/HPV16.mfa = the columns of all major-variant positions, reference row first, then
    one row per kept sample.  The expected rows are rebuilt here from the oracle's variant calls for each sample."""
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    paths, samples = [], []
    for i in range(4):
        gm, isnv = synth.sample_genome(g, 40 + i, n_snp=4 + i, n_isnv=2)
        if i == 3:
            gm = gm[:3000]                                   # a sample that covers less than 90 % of the genome
        reads = synth.codes_to_ascii(synth.single_end_codes(gm, 6000, 150, 140 + i, isnv=isnv if i < 3 else ()))
        p = str(tmp_path / ("smp%d.fastq.gz" % i))
        write_fastq_gz(p, reads, "s%d" % i)
        paths.append(p)
        samples.append(reads)
    out = str(tmp_path / "out")
    res = subprocess.run([BRONKO, "call", "-d", os.path.join(golden_dir, "hpv.bkdb"), "-r"] + paths + ["--alignment", "-o", out],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Skipping %s (breadth of coverage" % paths[3] in res.stdout
    # expected: from the oracle's calls
    cols, own = {}, []
    for i in range(4):
        pile = oracle.sample_pileup(ix, [samples[i]])
        recs, ptr, n, nmaj, nmin, br, dc = oracle.call_variants(ix, 0, pile, oracle.default_call_params(21))
        if br < 0.90:
            assert i == 3
            continue
        mine = {}
        for r in recs:
            if r["af"] >= 0.5:
                cols[int(r["pos"])] = "ACGT"[r["ref_base"]]
                mine[int(r["pos"])] = "ACGT"[r["alt_base"]]
        own.append((oracle.clean_sample_id(paths[i]), mine))
    assert len(own) == 3 and len(cols) >= 4
    order = sorted(cols)
    want = [">HPV16", "".join(cols[p] for p in order)]
    for name, mine in own:
        want += [">" + name, "".join(mine.get(p, cols[p]) for p in order)]
    got = open(os.path.join(out, "HPV16.mfa")).read().splitlines()
    assert got == want
    ix.close()


def test_call_three_samples_overlapped(oracle, golden_dir, tmp_path):
    """Several samples in one run: sample i+1 is ingested into a forked engine while a worker thread completes sample i
    (cli.cpp).  Every sample's VCF and pileup TSV must equal the oracle's, and bronko_overview.tsv keeps the input order."""
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    paths, samples = [], []
    for i in range(3):
        gm, isnv = synth.sample_genome(g, 70 + i, n_snp=3 + 2 * i, n_isnv=3)
        reads = synth.codes_to_ascii(synth.single_end_codes(gm, 7000 + 2500 * i, 150, 170 + i, isnv=isnv))
        p = str(tmp_path / ("multi%d.fastq.gz" % i))
        write_fastq_gz(p, reads, "m%d" % i)
        paths.append(p)
        samples.append(reads)
    out = str(tmp_path / "out")
    res = subprocess.run([BRONKO, "call", "-d", os.path.join(golden_dir, "hpv.bkdb"), "-r"] + paths + ["--pileup", "-o", out],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    odir = str(tmp_path / "oracle")
    os.makedirs(odir)
    ov = open(os.path.join(out, "bronko_overview.tsv")).read().splitlines()
    assert [line.split("\t")[0] for line in ov[1:]] == paths
    for i in range(3):
        stem, best, n, ov_want = oracle_outputs(oracle, ix, [samples[i]], odir, paths[i])
        for ext in (".vcf", ".tsv"):
            assert open(os.path.join(out, stem + ext), "rb").read() == open(os.path.join(odir, stem + ext), "rb").read(), (i, ext)
        perfect, variant, unmapped, nmaj, nmin, br, dc = ov_want
        assert ov[1 + i].split("\t")[2:] == [str(nmaj), str(nmin), "%.4f" % br, "%.4f" % dc, str(perfect), str(variant), str(unmapped)]
    ix.close()


@pytest.mark.parametrize("lane_env,n_lanes", [({"BRONKO_DEVICES": "0,0"}, 2), ({"BRONKO_LANES": "3"}, 3)])
def test_call_samples_dealt_to_several_lanes_on_the_one_gpu_of_the_box(oracle, golden_dir, tmp_path, lane_env, n_lanes):
    """Whole samples per GPU, no collective (call.rs:212: samples are independent): `bronko call` deals the samples to lanes in
    turn -- host threads that ingest into their own engine (+ fork); the lanes of one device share its tables.
    BRONKO_DEVICES=0,0 runs two lanes on the one GPU of the test box (the code path of two GPUs; a second physical device has
    never been available to these tests), BRONKO_LANES=3 three lanes on it (what -t 6 gives): five samples, every output equal to
    the oracle's, overview in input order."""
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    paths, samples = [], []
    for i in range(5):
        gm, isnv = synth.sample_genome(g, 80 + i, n_snp=2 + i, n_isnv=3)
        reads = synth.codes_to_ascii(synth.single_end_codes(gm, 5000 + 1500 * i, 150, 180 + i, isnv=isnv))
        p = str(tmp_path / ("lane%d.fastq.gz" % i))
        write_fastq_gz(p, reads, "l%d" % i)
        paths.append(p)
        samples.append(reads)
    out = str(tmp_path / "out")
    env = dict(os.environ, **lane_env)
    res = subprocess.run([BRONKO, "call", "-d", os.path.join(golden_dir, "hpv.bkdb"), "-r"] + paths + ["--pileup", "-o", out],
                         capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "%d GPU lanes" % n_lanes in res.stdout + res.stderr
    odir = str(tmp_path / "oracle")
    os.makedirs(odir)
    ov = open(os.path.join(out, "bronko_overview.tsv")).read().splitlines()
    assert [line.split("\t")[0] for line in ov[1:]] == paths
    for i in range(5):
        stem, best, n, ov_want = oracle_outputs(oracle, ix, [samples[i]], odir, paths[i])
        for ext in (".vcf", ".tsv"):
            assert open(os.path.join(out, stem + ext), "rb").read() == open(os.path.join(odir, stem + ext), "rb").read(), (i, ext)
    ix.close()


def test_call_one_sample_sharded_through_rccl(oracle, golden_dir, sars_paths, tmp_path):
    """BRONKO_SHARD=1: a sample's batches are dealt to one engine per GPU and the counter planes are reduce-scattered by RCCL from
    inside the binary (cli.cpp sharded_finalize: the k-mer statistics tables' exchange, bk_shard_measure / _transport /
    ncclReduceScatter / _received per mate file, bk_sample_finalize_shard, three all-reduces, bk_sample_merge_shards).  On this
    box that is a communicator of ONE rank -- every collective still runs through the backend -- and the outputs must be the
    unsharded run's and the oracle's, byte for byte: a paired HPV16 sample, and a four-strain sample (every genome's rows: the
    selected-only finalize cannot be sharded)."""
    env = dict(os.environ, BRONKO_SHARD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # paired HPV16
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    gm, isnv = synth.sample_genome(g, 21, n_snp=5, n_isnv=5)
    c1, c2 = synth.paired_codes(gm, 12000, 150, 21, isnv=isnv)
    r1, r2 = synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)
    r1[5] = r1[5][:33] + b"N" + r1[5][34:]
    p1, p2 = str(tmp_path / "shard_R1.fastq.gz"), str(tmp_path / "shard_R2.fastq.gz")
    write_fastq_gz(p1, r1, "r1")
    write_fastq_gz(p2, r2, "r2")
    outs = {}
    for name, e in (("sharded", env), ("plain", dict(os.environ, BRONKO_SHARD="0"))):
        out = str(tmp_path / ("out_" + name))
        res = subprocess.run([BRONKO, "call", "-d", os.path.join(golden_dir, "hpv.bkdb"), "-1", p1, "-2", p2, "--pileup", "-o", out, "-t", "2"],
                             capture_output=True, text=True, env=e)
        assert res.returncode == 0, res.stdout + res.stderr
        assert ("RCCL reduce-scatter" in res.stdout) == (name == "sharded"), res.stdout
        outs[name] = out
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    odir = str(tmp_path / "oracle")
    os.makedirs(odir)
    stem, best, n, ov_want = oracle_outputs(oracle, ix, [r1, r2], odir, p1)
    for ext in (".vcf", ".tsv"):
        want = open(os.path.join(odir, stem + ext), "rb").read()
        assert open(os.path.join(outs["sharded"], stem + ext), "rb").read() == want, ext
        assert open(os.path.join(outs["plain"], stem + ext), "rb").read() == want, ext
    ov = [open(os.path.join(outs[nm], "bronko_overview.tsv")).read() for nm in ("sharded", "plain")]
    assert ov[0] == ov[1]   # incl. num_unmapped_kmers: KMC's counted k-mers survive the tables' exchange
    ix.close()
    # four strains, single-end
    g = synth.read_fasta_bytes(sars_paths[2])
    gm, isnv = synth.sample_genome(g, 22)
    reads = synth.codes_to_ascii(synth.single_end_codes(gm, 20000, 150, 22, isnv=isnv))
    fq = str(tmp_path / "s4.fq")
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@s_%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    out = str(tmp_path / "o4")
    res = subprocess.run([BRONKO, "call", "-g"] + sars_paths + ["-r", fq, "--pileup", "-o", out, "-t", "2"], capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    ix = oracle.Index.build(21, sars_paths)
    odir = str(tmp_path / "oracle4")
    os.makedirs(odir)
    stem, best, n, ov_want = oracle_outputs(oracle, ix, [reads], odir, fq)
    for ext in (".vcf", ".tsv"):
        assert open(os.path.join(out, stem + ext), "rb").read() == open(os.path.join(odir, stem + ext), "rb").read(), ext
    ix.close()


def test_call_reads_inflated_on_several_threads(golden_dir, tmp_path):
    """`-t 16` for one paired sample gives each of the two files eight inflate threads (bronko_amd/host/pargz.hpp; KMC reads the
    reference's input with -t threads, call.rs:1166-1181): every output file is the one the single-thread reader produces."""
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    gm, isnv = synth.sample_genome(g, 4, n_snp=6, n_isnv=6)
    c1, c2 = synth.paired_codes(gm, 60000, 150, 4, isnv=isnv)
    r1, r2 = synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)
    p1, p2 = str(tmp_path / "s_R1.fastq.gz"), str(tmp_path / "s_R2.fastq.gz")
    rng = np.random.default_rng(4)
    for p, rs, tag in ((p1, r1, "a"), (p2, r2, "b")):
        with gzip.open(p, "wb", compresslevel=6) as f:
            for i, r in enumerate(rs):
                f.write(b"@%s_%d\n%s\n+\n%s\n" % (tag.encode(), i, r, (rng.integers(0, 8, len(r)) * 5 + 35).astype(np.uint8).tobytes()))
    outs = []
    for name, env in (("one", {"BRONKO_INFLATE_THREADS": "1"}), ("many", {}), ("in_turn", {"BRONKO_NO_READ_AHEAD": "1"})):   # (in_turn: files opened when their sample's turn comes)
        out = str(tmp_path / name)
        res = subprocess.run([BRONKO, "call", "-d", os.path.join(golden_dir, "hpv.bkdb"), "-1", p1, "-2", p2, "--pileup", "-o", out, "-t", "16"],
                             capture_output=True, text=True, env={**os.environ, **env})
        assert res.returncode == 0, res.stdout + res.stderr
        assert ("inflated on 8 threads" in res.stdout + res.stderr) == (name != "one")
        outs.append({f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))})
    assert outs[0].keys() == outs[1].keys() == outs[2].keys() and len(outs[0]) >= 3
    for f in outs[0]:
        assert outs[0][f] == outs[1][f] == outs[2][f], f
