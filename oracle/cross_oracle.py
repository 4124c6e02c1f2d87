#!/usr/bin/env python3
"""cross_oracle.py -- a SECOND, independent restatement of bronko's `call` k-mer -> pileup path (test infrastructure only).

Written from the Rust text of /root/reference alone (not from oracle/bronko_oracle.c), in plain Python with dictionaries,
strings and arbitrary-size integers masked to 64 bits -- structurally as far from the C oracle as the reference allows:

    lcb.rs:1-45     assign_buckets          lcb.rs:47-55   nt_to_bits        lcb.rs:67-74  kmer_to_u64
    lcb.rs:76-85    reverse_complement_u64  lcb.rs:87-95   canonical_kmer
    build.rs:145-231 build_indexes  (dict bucket id -> list of (file_id, seq_id, location, idx, canonical))
    call.rs:1152-1255 the KMC3 run + dump + load_kmers: KMC3 is not under /root/reference; its documented contract for
                    `-k -b -ci -cs1000000` is restated here as count_kmers -- the one part that is not Rust text
    call.rs:1257-1434 map_kmers (k-mers travel as STRINGS, like upstream's Vec<(String, u64)>)
    call.rs:1437-1480 initialize_output_maps (nested lists [pos][base] per (file, sequence name))

Run in the build container, it writes small known-answer fixtures to tests/golden/call_*.npz; tests/test_cross_oracle.py asserts
that the C oracle reproduces them.  Two restatements written separately agreeing on every cell does not pin parity to upstream
(nothing can without cargo + KMC), but it halves the chance that the GPU path and its checker share one misreading.

usage: python oracle/cross_oracle.py            (re)generate tests/golden/call_{hpv_snp,hpv_reads,sars4_pairs,sars2_k31}.npz
       python oracle/cross_oracle.py --fuzz N   (re)generate tests/golden/call_fuzz.npz: N (default 200) small fuzz-shaped cases
"""
import os
import sys

M64 = (1 << 64) - 1


def nt_to_bits(nt):                                   # lcb.rs:47-55
    return {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}.get(nt, 0)


def kmer_to_u64(kmer):                                # lcb.rs:67-74
    val = 0
    for base in kmer:
        val = ((val << 2) & M64) | nt_to_bits(base)
    return val


def reverse_complement_u64(kmer_val, k):              # lcb.rs:76-85
    rc = 0
    for i in range(k):
        two_bits = (kmer_val >> (2 * i)) & 0b11
        rc = ((rc << 2) & M64) | (0b11 ^ two_bits)
    return rc


def canonical_kmer(kmer, k):                          # lcb.rs:87-95
    fwd = kmer_to_u64(kmer)
    rev = reverse_complement_u64(fwd, k)
    return (fwd, False) if fwd < rev else (rev, True)


def assign_buckets(kmer, k):                          # lcb.rs:1-45 (u64 arithmetic wraps)
    buckets, num_a, val, mu = [0] * k, [0] * k, [0] * k, [0] * k
    mask = (3 << ((k - 1) * 2)) & M64
    p = (1 << ((k - 1) * 2)) & M64
    cur = kmer & mask
    val[0] = (kmer - cur) & M64
    mu[0] = (p + ((cur >> 2) * (k - 1))) & M64 if cur != 0 else val[0]
    sum_mu = mu[0]
    for i in range(1, k):
        num_a[i] = num_a[i - 1] + (1 if cur == 0 else 0)
        mask >>= 2
        cur = kmer & mask
        p >>= 2
        val[i] = (val[i - 1] - cur) & M64
        mu[i] = (p + ((cur >> 2) * (k - i - 1))) & M64 if cur != 0 else val[i]
        sum_mu = (sum_mu + mu[i]) & M64
    mask = (3 << ((k - 1) * 2)) & M64
    for i in range(k):
        cur = kmer & mask
        mask >>= 2
        buckets[i] = (sum_mu - mu[i] + val[i] - num_a[i] * cur + 1 + num_a[i]) & M64
    return buckets


def read_fasta(path):
    """needletail's view of a plain FASTA file: (id line without '>', sequence with line breaks removed)"""
    recs, name, seq = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(seq)))
                name, seq = line[1:], []
            elif name is not None:
                seq.append(line)
    if name is not None:
        recs.append((name, b"".join(seq)))
    return recs


def build_indexes_files(k, files_in):                  # build.rs:145-231 on records already read: [(file stem, [(id line, sequence)])]
    global_index, files = {}, []
    for file_id, (file_name, records) in enumerate(files_in):
        sequences = []
        for seq_id, (rid, seq) in enumerate(records):
            parts = rid.decode("utf-8", "replace").split()
            sequences.append((parts[0] if parts else "", len(seq), seq))
            for i in range(0, max(len(seq) - k, 0) + 1):                       # 0..=seq_len.saturating_sub(k)
                kmer = seq[i:i + k]
                if len(kmer) < k:
                    raise IndexError("slice out of range (upstream panics on a sequence shorter than k)")
                kmer_bin, canonical = canonical_kmer(kmer, k)
                for j, bucket_id in enumerate(assign_buckets(kmer_bin, k)):
                    global_index.setdefault(bucket_id, []).append((file_id, seq_id, i, j, canonical))
        files.append((file_name, sequences))
    return global_index, files


def build_indexes(k, genomes):                        # build.rs:145-231 (files through needletail's eyes)
    return build_indexes_files(k, [(os.path.splitext(os.path.basename(p))[0], read_fasta(p)) for p in genomes])   # Path::file_stem


def count_kmers(reads, k, ci, cs=1000000, cx=1000000000):
    """KMC3 contract for `kmc -k{k} -b -ci{ci} -cs1000000` + `kmc_tools transform dump` (call.rs:1166-1177, :1203-1211):
    every window of k consecutive ACGT/acgt symbols of a read is one occurrence of that k-mer, counted as it stands on the
    read strand (-b); any other symbol breaks the run; k-mers seen fewer than ci times (or more than cx) are dropped; the
    stored count saturates at cs.  Returns ([(k-mer string, count)], (total reads, total k-mers, unique, unique counted))."""
    counts = {}
    total_kmers = 0
    for read in reads:
        run = bytearray()
        for sym in read:
            if sym in b"ACGTacgt":
                run.append(sym)
            else:
                run = bytearray()
                continue
            if len(run) >= k:
                key = bytes(run[-k:]).upper()
                counts[key] = counts.get(key, 0) + 1
                total_kmers += 1
    kept = [(key.decode(), min(n, cs)) for key, n in counts.items() if ci <= n <= cx]
    return kept, (len(reads), total_kmers, len(counts), len(kept))


def initialize_output_maps(files):                    # call.rs:1437-1480
    def one():
        return {fid: {name: [[0, 0, 0, 0] for _ in range(ln)] for (name, ln, _) in seqs} for fid, (_, seqs) in enumerate(files)}
    return one(), one(), one(), one()                 # output, output_rev, output_counts, output_rev_counts


def map_kmers(kmers, index, files, k, n_fixed, use_full_kmer, output_maps):   # call.rs:1257-1434
    output, output_rev, output_counts, output_rev_counts = output_maps
    results = {}
    for kmer, n in kmers:
        kmer_bin, rc = canonical_kmer(kmer.encode(), k)
        buckets = assign_buckets(kmer_bin, k)
        if use_full_kmer:
            filtered = buckets
        elif n_fixed * 2 + 1 >= len(buckets):
            filtered = []
        else:
            filtered = buckets[n_fixed:len(buckets) - n_fixed - 1]
        num_buckets_perfect = len(filtered)
        hits = {}
        for bucket in filtered:
            for (file_id, seq_id, location, idx_in, canonical) in index.get(bucket, ()):
                hits[file_id] = hits.get(file_id, 0) + 1
                seq = files[file_id][1][seq_id][0]
                nuc_x = idx_in
                if canonical:
                    pos = k - nuc_x - 1
                    bit_idx = ((kmer_bin >> (2 * (k - pos - 1))) & 0b11) ^ 0b11
                    idx = location + nuc_x
                    cnt, dep = (output_counts, output) if rc else (output_rev_counts, output_rev)
                else:
                    pos = nuc_x
                    bit_idx = (kmer_bin >> (2 * (k - pos - 1))) & 0b11
                    idx = location + nuc_x
                    cnt, dep = (output_rev_counts, output_rev) if rc else (output_counts, output)
                cnt[file_id][seq][idx][bit_idx] += 1
                if dep[file_id][seq][idx][bit_idx] < n:
                    dep[file_id][seq][idx][bit_idx] = n
        perfect = [f for f, h in hits.items() if h == num_buckets_perfect]
        for f, h in hits.items():
            e = results.setdefault(f, [0, 0, 0])
            if h == num_buckets_perfect:
                e[0] += 1
            elif h > 0:
                e[1] += 1
        if len(perfect) == 1:
            results.setdefault(perfect[0], [0, 0, 0])[2] += 1
    return results


def sample(genomes, k, mates, ci=3, n_fixed=2, use_full_kmer=False, in_memory=False):
    """One sample the way call.rs:298-317 runs it: one KMC run per mate file, R1 then R2 mapped into the same arrays.
    Returns flat numpy arrays in (file, sequence, position, base) order + per-mate stats / kmc stats.
    in_memory: genomes = [(file stem, [(id line, sequence)])] instead of FASTA paths."""
    import numpy as np
    index, files = build_indexes_files(k, genomes) if in_memory else build_indexes(k, genomes)
    maps = initialize_output_maps(files)
    stats = np.zeros((len(mates), len(files), 3), np.uint64)
    present = np.zeros((len(mates), len(files)), np.uint8)
    kmc = np.zeros((len(mates), 4), np.uint64)
    for m, reads in enumerate(mates):
        kmers, st = count_kmers(reads, k, ci)
        kmc[m] = st
        for f, e in map_kmers(kmers, index, files, k, n_fixed, use_full_kmer, maps).items():
            stats[m, f] = e
            present[m, f] = 1
    flat = []
    for mp in maps:
        rows = [row for fid, (_, seqs) in enumerate(files) for (name, _, _) in seqs for row in mp[fid][name]]
        flat.append(np.array(rows, np.uint64).reshape(-1))
    return flat, stats, present, kmc


def fuzz_case(np, seed):
    """One small case in the shape of tools/fuzz_parity.py's: a random genome with direct / reverse-complement repeats and
    low-complexity stretches, 1-8 files of 1-3 sequences derived from it by substitutions, k from 11 to 31, the window variants
    (n_fixed 0 / 1 / 2 / 5 incl. the empty window, --use-full-kmer), ci 1-3, reads of 20-200 bases with substitutions, indels,
    chimeras, foreign reads, N and lower case, on either strand, one or two mate files."""
    rng = np.random.default_rng(seed)
    B = b"ACGT"
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rand_seq = lambda n: bytes(B[i] for i in rng.integers(0, 4, n))
    k = int(rng.choice([11, 15, 19, 21, 21, 21, 25, 31, 31]))
    n = int(rng.integers(max(120, 4 * k), 900))
    g = bytearray(rand_seq(n))
    for _ in range(int(rng.integers(0, 3))):
        ln = int(rng.integers(15, 80)); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln))
        seg = bytes(g[a:a + ln])
        if rng.random() < 0.5: seg = seg.translate(comp)[::-1]
        g[b:b + ln] = seg
    if rng.random() < 0.4:
        ln = int(rng.integers(10, 50)); a = int(rng.integers(0, n - ln))
        unit = rand_seq(int(rng.integers(1, 3)))
        g[a:a + ln] = (unit * ln)[:ln]
    base = bytes(g)

    def mutate(x, n_sub):
        x = bytearray(x)
        for p in rng.integers(0, len(x), n_sub):
            x[p] = B[(B.index(x[p]) + int(rng.integers(1, 4))) & 3]
        return bytes(x)

    files = []
    for f in range(int(rng.choice([1, 1, 2, 2, 3, 4, 8]))):
        x = base if f == 0 else mutate(base, int(rng.integers(0, 12)))
        cuts = sorted(set([0, len(x)] + [int(c) for c in rng.integers(k, len(x) - k, int(rng.integers(0, 3)))]))
        seqs = [(("s%d_%d comment" % (f, i)).encode(), x[a:b]) for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])) if b - a >= k]
        if seqs:
            files.append(("file%d" % f, seqs))
    n_fixed = int(rng.choice([2, 2, 2, 0, 1, 5]))
    full = bool(rng.random() < 0.15)
    ci = int(rng.choice([1, 1, 2, 3]))
    src = mutate(base, int(rng.integers(0, 6)))
    err = float(rng.choice([0.0, 0.005, 0.02, 0.05]))
    reads = []
    for _ in range(int(rng.integers(4, 90))):
        ln = int(rng.integers(20, min(200, len(src))))
        a = int(rng.integers(0, len(src) - ln + 1))
        r = bytearray(src[a:a + ln])
        u = rng.random()
        if u < 0.05: r = bytearray(rand_seq(ln))
        elif u < 0.10 and ln > 60:
            b2 = int(rng.integers(0, len(src) - ln + 1)); r[ln // 2:] = src[b2 + ln // 2:b2 + ln]
        elif u < 0.15 and ln > 40:
            p = int(rng.integers(10, ln - 10))
            r = r[:p] + (bytearray(rand_seq(int(rng.integers(1, 4)))) if rng.random() < 0.5 else bytearray()) + r[p + int(rng.integers(0, 4)):]
        for p in np.nonzero(rng.random(len(r)) < err)[0]:
            r[p] = B[(B.index(r[p]) + int(rng.integers(1, 4))) & 3]
        if rng.random() < 0.04 and len(r) > 5: r[int(rng.integers(0, len(r)))] = ord("N")
        r = bytes(r)
        if rng.random() < 0.5: r = r.translate(comp)[::-1]
        if rng.random() < 0.05: r = r.lower()
        # (a k-mer seen several times: ci > 1 must not leave everything below the threshold)
        reads.extend([r] * (int(rng.integers(1, 4)) if ci > 1 else 1))
    mates = [reads]
    if rng.random() < 0.3 and len(reads) > 1:
        h = len(reads) // 2
        mates = [reads[:h], reads[h:]]
    return files, k, mates, dict(ci=ci, n_fixed=n_fixed, use_full_kmer=full)


def write_fuzz(n_cases, seed0=20261003):
    """tests/golden/call_fuzz.npz: n_cases small cases (fuzz_case), genomes and reads inside the fixture, results of THIS restatement."""
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {"n_cases": n_cases}
    for c in range(n_cases):
        files, k, mates, kw = fuzz_case(np, seed0 + c)
        flat, stats, present, kmc = sample(files, k, mates, in_memory=True, **kw)
        nz = [np.nonzero(a)[0].astype(np.uint32) for a in flat]
        pre = "c%03d_" % c
        text = b"".join(b"F\t" + fn.encode() + b"\n" + b"".join(b"S\t" + rid + b"\t" + sq + b"\n" for rid, sq in seqs) for fn, seqs in files)
        out.update({pre + "files": np.frombuffer(text, np.uint8), pre + "k": k, pre + "n_fixed": kw["n_fixed"], pre + "full": int(kw["use_full_kmer"]),
                    pre + "ci": kw["ci"], pre + "n_mates": len(mates), pre + "reads0": np.frombuffer(b"\n".join(mates[0]), np.uint8),
                    pre + "reads1": np.frombuffer(b"\n".join(mates[1]) if len(mates) > 1 else b"", np.uint8), pre + "n_cells4": len(flat[0]),
                    pre + "stats": stats, pre + "present": present, pre + "kmc": kmc})
        for name, a, z in zip(("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"), flat, nz):
            out[pre + name + "_idx"], out[pre + name + "_val"] = z, a[z]
        print("case %3d k=%2d n_fixed=%d full=%d ci=%d files=%d cells=%d reads=%s: non-zero cells %s, stats %s" %
              (c, k, kw["n_fixed"], kw["use_full_kmer"], kw["ci"], len(files), len(flat[0]) // 4, [len(m) for m in mates], [len(x) for x in nz],
               stats.sum(axis=0).tolist()[:3]), flush=True)
    np.savez_compressed(os.path.join(root, "tests", "golden", "call_fuzz.npz"), **out)


def main():
    import numpy as np
    if len(sys.argv) > 1 and sys.argv[1] == "--fuzz":
        return write_fuzz(int(sys.argv[2]) if len(sys.argv) > 2 else 200)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bronko_amd import synth   # (inputs only: the seeded read generator)
    golden = os.path.join(root, "tests", "golden")
    sars = [os.path.join(golden, "4_sarscov2", n) for n in ("wuhan_ref.fasta", "OM223929.1.fasta", "ON765678.1.fasta", "PX392231.1.fasta")]
    hpv = os.path.join(golden, "HPV16.fa")
    cases = []
    # 1. the derived known answer of SURVEY.md §8c: HPV16 k=21, SNP A->T at 0-based 1000, the 21 covering windows 10x + their
    #    reverse complements 7x -- given as reads
    g = bytearray(synth.read_fasta_bytes(hpv))
    g[1000] = ord("T")
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    wins = [bytes(g[s:s + 21]) for s in range(980, 1001)]
    reads = [w for w in wins for _ in range(10)] + [w.translate(comp)[::-1] for w in wins for _ in range(7)]
    cases.append(("hpv_snp", [hpv], 21, [reads], dict(ci=3)))
    # 2. HPV16, seeded 150 bp reads with errors, iSNVs, N symbols and short reads
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(hpv), 41)
    rd = synth.codes_to_ascii(synth.single_end_codes(gm, 2500, 150, 41, err=0.01, isnv=isnv))
    rd[3] = rd[3][:70] + b"N" + rd[3][71:]
    rd[9] = b"ACGTACGT"
    rd[11] = rd[11].lower()
    cases.append(("hpv_reads", [hpv], 21, [rd], dict(ci=2)))
    # 3. four SARS-CoV-2 strains, paired-end sample derived from ON765678.1 (selection + shared buckets)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars[2]), 43)
    c1, c2 = synth.paired_codes(gm, 1500, 150, 43, isnv=isnv)
    cases.append(("sars4_pairs", sars, 21, [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)], dict(ci=1)))
    # 4. k = 31 (bucket ids wrap modulo 2^64), --use-full-kmer off / n_fixed 3, two strains
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars[1]), 47)
    rd = synth.codes_to_ascii(synth.single_end_codes(gm, 1200, 150, 47, err=0.01, isnv=isnv))
    cases.append(("sars2_k31", sars[:2], 31, [rd], dict(ci=1, n_fixed=3)))
    for name, genomes, k, mates, kw in cases:
        flat, stats, present, kmc = sample(genomes, k, mates, **kw)
        nz = [np.nonzero(a)[0].astype(np.uint32) for a in flat]
        np.savez_compressed(os.path.join(golden, "call_%s.npz" % name),
                            genomes=np.array([os.path.relpath(p, golden) for p in genomes]), k=k,
                            n_fixed=kw.get("n_fixed", 2), ci=kw.get("ci", 3), n_mates=len(mates),
                            reads0=np.frombuffer(b"\n".join(mates[0]), np.uint8),
                            reads1=np.frombuffer(b"\n".join(mates[1]) if len(mates) > 1 else b"", np.uint8), n_cells4=len(flat[0]),
                            fwd_depth_idx=nz[0], fwd_depth_val=flat[0][nz[0]], rev_depth_idx=nz[1], rev_depth_val=flat[1][nz[1]],
                            fwd_nk_idx=nz[2], fwd_nk_val=flat[2][nz[2]], rev_nk_idx=nz[3], rev_nk_val=flat[3][nz[3]],
                            stats=stats, present=present, kmc=kmc)
        print("%-12s k=%d mates=%d reads=%d: non-zero cells %s, stats %s" % (name, k, len(mates), sum(len(m) for m in mates), [len(x) for x in nz], stats.sum(axis=0).tolist()))


if __name__ == "__main__":
    main()
