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

usage: python oracle/cross_oracle.py            (re)generate tests/golden/call_*.npz
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


def build_indexes(k, genomes):                        # build.rs:145-231
    global_index, files = {}, []
    for file_id, file_path in enumerate(genomes):
        file_name = os.path.splitext(os.path.basename(file_path))[0]          # Path::file_stem
        sequences = []
        for seq_id, (rid, seq) in enumerate(read_fasta(file_path)):
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


def sample(genomes, k, mates, ci=3, n_fixed=2, use_full_kmer=False):
    """One sample the way call.rs:298-317 runs it: one KMC run per mate file, R1 then R2 mapped into the same arrays.
    Returns flat numpy arrays in (file, sequence, position, base) order + per-mate stats / kmc stats."""
    import numpy as np
    index, files = build_indexes(k, genomes)
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


def main():
    import numpy as np
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
