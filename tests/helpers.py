"""Shared helpers for the parity tests (tests/ only)."""
import os

import numpy as np

from bronko_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def engine_from_oracle_index(ix, params=None):
    """Feed the HIP engine with the flattened arrays of an oracle-held index (same data a .bkdb decodes to)."""
    from bronko_amd import Engine
    return Engine(ix.k, ix.bucket_ids(), ix.bucket_off(), ix.entries(), ix.files(), params)


def hip_sample(eng, mates_ascii, k, stride_words=None, batch=None, ascii_path=False):
    """Run one sample through the C ABI: pack (K0, host or -- ascii_path -- on the GPU) -> push per mate -> finish."""
    from bronko_amd import pack_reads
    eng.sample_begin()
    if ascii_path:
        for m, reads in enumerate(mates_ascii):
            step = batch or max(len(reads), 1)
            for i in range(0, len(reads), step):
                eng.push_reads_ascii(m, reads[i:i + step])
        return eng.sample_finish(len(mates_ascii))
    for m, reads in enumerate(mates_ascii):
        words, lens = pack_reads(reads, k, stride_words)
        if batch:
            for i in range(0, len(lens), batch):
                eng.push_reads(m, words[i:i + batch], lens[i:i + batch])
        else:
            eng.push_reads(m, words, lens)
    return eng.sample_finish(len(mates_ascii))


def sharded_finalize_on_one_device(engs, n_mates, width):
    """The sharded finalize of include/bronko_hip.h with len(engs) engines on ONE device standing for as many ranks, each of
    which has scanned its share of the sample: bk_shard_transport packs every plane, the reduce-scatter is simulated (typed,
    wrapping sums of the send buffers; part r to rank r), bk_shard_received widens, bk_sample_finalize_shard maps, the small
    results are combined (max / sum) and installed on every rank.  Afterwards every engine's sample_download is the sample's."""
    import torch
    from bronko_amd.dist import DeviceVector
    world = len(engs)
    item, tstr = (8, "<i8") if width == 64 else (4, "<i4")
    cells4 = engs[0].total_cells * 4
    for m in range(n_mates):
        bufs = []
        for e in engs:
            sp, pb, rp = e.shard_transport(m, world, width)
            bufs.append((torch.as_tensor(DeviceVector(sp, pb // item * world, tstr), device="cuda:0"),
                         torch.as_tensor(DeviceVector(rp, pb // item, tstr), device="cuda:0")))
        torch.cuda.synchronize()
        total = bufs[0][0].clone()
        for send, _ in bufs[1:]:
            total += send                                                  # (wraps like the collective's sum)
        n = bufs[0][1].numel()
        for r, (_, recv) in enumerate(bufs):
            recv.copy_(total[r * n:(r + 1) * n])
        torch.cuda.synchronize()
        for r, e in enumerate(engs):
            e.shard_received(m, r, world, width)
    piles, sums = [], []
    for r, e in enumerate(engs):
        e.sample_finalize_shard(n_mates, r, world)
        piles.append(torch.as_tensor(DeviceVector(e.pileup_ptr(), 4 * cells4), device="cuda:0"))
        sp, sn = e.shard_sums()
        sums.append(torch.as_tensor(DeviceVector(sp, sn), device="cuda:0"))
    torch.cuda.synchronize()
    depth = torch.stack([p[:2 * cells4] for p in piles]).max(dim=0).values
    nk = torch.stack([p[2 * cells4:] for p in piles]).sum(dim=0)
    ssum = torch.stack(sums).sum(dim=0)
    for r, e in enumerate(engs):
        piles[r][:2 * cells4] = depth
        piles[r][2 * cells4:] = nk
        sums[r].copy_(ssum)
    torch.cuda.synchronize()
    for e in engs:
        e.sample_merge_shards()


def assert_same_pileup(res, pile):
    for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
        a, b = getattr(res, name), getattr(pile, name)
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError("%s differs in %d of %d cells; first at %d: hip=%d oracle=%d" %
                                 (name, len(bad), len(a), bad[0], a[bad[0]], b[bad[0]]))
    assert np.array_equal(res.stats, pile.stats), (res.stats, pile.stats)
    assert np.array_equal(res.present, pile.present)


def hpv_reads(n, seed, read_len=150, err=0.005, with_n=False, ragged=False):
    g = synth.read_fasta_bytes(os.path.join(GOLDEN, "HPV16.fa"))
    gm, isnv = synth.sample_genome(g, seed)
    codes = synth.single_end_codes(gm, n, read_len, seed + 100, err=err, isnv=isnv)
    reads = synth.codes_to_ascii(codes)
    if with_n or ragged:
        r = synth.splitmix64(seed + 999, 3 * n)
        out = []
        for i, rd in enumerate(reads):
            rd = bytearray(rd)
            if with_n and r[3 * i] % np.uint64(10) == 0:
                rd[int(r[3 * i + 1] % np.uint64(len(rd)))] = ord("N")
            if ragged:
                rd = rd[: int(r[3 * i + 2] % np.uint64(len(rd) + 1))]
            out.append(bytes(rd))
        reads = out
    return reads


# ---- the committed fixtures of the second restatement (oracle/cross_oracle.py) ------------------------------------------------
def expand_sparse(z, name, pre=""):
    a = np.zeros(int(z[pre + "n_cells4"]), np.uint64)
    a[z[pre + name + "_idx"]] = z[pre + name + "_val"]
    return a


def fuzz_fixture_case(z, c):
    """Case c of tests/golden/call_fuzz.npz: (files [(file name, [(sequence id line, sequence)])], k, mates, keyword arguments, prefix)."""
    pre = "c%03d_" % c
    files = []
    for line in bytes(z[pre + "files"]).split(b"\n"):
        if line.startswith(b"F\t"):
            files.append((line[2:].decode(), []))
        elif line.startswith(b"S\t"):
            rid, seq = line[2:].split(b"\t")
            files[-1][1].append((rid, seq))
    n_mates = int(z[pre + "n_mates"])
    mates = [bytes(z[pre + "reads%d" % m]).split(b"\n") for m in range(n_mates)]
    kw = dict(ci=int(z[pre + "ci"]), n_fixed=int(z[pre + "n_fixed"]), use_full_kmer=bool(int(z[pre + "full"])))
    return files, int(z[pre + "k"]), mates, kw, pre


# ---- exact (unwrapped) LCB bucket ranks: test-side restatement used to construct k = 31 aliasing reads ----------
def lcb_rank(v, pos, k):
    """1-based lexicographic rank of (v, pos) among all (k-mer with an A at `pos`, position) pairs, ordered by k-mer
    value then position; v must have A (0) at `pos`.  lcb.rs:1-45 computes this modulo 2^64 (checked on CPU in
    tests/test_oracle_golden.py::test_exact_rank_matches_assign_buckets)."""
    cum, a_pre = 0, 0
    for i in range(k):
        d = (v >> (2 * (k - 1 - i))) & 3
        rest = k - 1 - i
        pw = 4 ** rest
        free_a = rest * (pw // 4) if rest else 0
        for x in range(d):
            cum += (a_pre + (x == 0)) * pw + free_a
        a_pre += d == 0
    before = sum(1 for i in range(pos) if ((v >> (2 * (k - 1 - i))) & 3) == 0)
    return cum + before + 1


def lcb_unrank(r1, k):
    """Inverse of lcb_rank: (v, pos) or None when r1 is not a rank."""
    if r1 <= 0:
        return None
    r, v, a_pre = r1 - 1, 0, 0
    for i in range(k):
        rest = k - 1 - i
        pw = 4 ** rest
        free_a = rest * (pw // 4) if rest else 0
        for x in range(4):
            c = (a_pre + (x == 0)) * pw + free_a
            if r < c:
                break
            r -= c
        else:
            return None
        v |= x << (2 * rest)
        a_pre += x == 0
    for i in range(k):
        if ((v >> (2 * (k - 1 - i))) & 3) == 0:
            if r == 0:
                return v, i
            r -= 1
    return None


def kmer_str(v, k):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))
