"""Seeded synthetic reads: splitmix64 streams, identical for every consumer of a seed.

Read model: start ~ U[0, L-len]; strand ~ Bernoulli(0.5) (single-end); per-base substitution error `err`
uniform over the 3 other bases; the sample genome is the reference plus fixed SNPs (AF 1.0) and iSNVs
(AF ~ U[0.03, 0.3]).  Paired-end: fragment of `frag_len`, R1 = its first read_len bases, R2 = reverse
complement of its last read_len bases.  Used by tests and bench.py (inputs only; no compute of the path).
"""
import numpy as np

_M = (1 << 64) - 1
BASES = np.frombuffer(b"ACGT", np.uint8)
CODE = np.full(256, 0, np.uint8)  # non-ACGT -> 0 like lcb.rs:53
for _i, _c in enumerate(b"ACGT"):
    CODE[_c] = _i
    CODE[_c + 32] = _i


def splitmix64(seed, n):
    """n outputs of the splitmix64 stream started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def read_fasta_bytes(path):
    """Concatenated sequence of the first record of a plain FASTA file (fixture helper)."""
    seq = []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if seq:
                    break
                continue
            seq.append(line.strip())
    return b"".join(seq)


def sample_genome(ref, seed, n_snp=20, n_isnv=20):
    """Returns (major genome bytes, [(pos, alt_base_code, af), ...] iSNVs)."""
    g = np.frombuffer(bytes(ref), np.uint8).copy()
    L = len(g)
    r = splitmix64(seed ^ 0xA11CE, 4 * (n_snp + n_isnv))
    pos = (r[0::4] % np.uint64(L)).astype(np.int64)
    shift = (r[1::4] % np.uint64(3)).astype(np.int64) + 1
    af = 0.03 + 0.27 * ((r[2::4] >> np.uint64(11)).astype(np.float64) / (1 << 53))
    isnv = []
    for i in range(n_snp + n_isnv):
        p = int(pos[i])
        alt = (int(CODE[g[p]]) + int(shift[i])) & 3
        if i < n_snp:
            g[p] = BASES[alt]
        else:
            isnv.append((p, alt, float(af[i])))
    return g.tobytes(), isnv


def _apply_isnv(codes, start, span, isnv, seed, n):
    for j, (p, alt, af) in enumerate(isnv):  # each covering read carries the alt with probability af
        rows = np.nonzero((start <= p) & (p < start + span))[0]
        if len(rows) == 0:
            continue
        u = _uniform(seed ^ (0x15A70000 + j), n)[rows]
        rows = rows[u < af]
        codes[rows, p - start[rows]] = alt


def _apply_errors(codes, seed, err):
    if err <= 0:
        return codes
    n, ln = codes.shape
    u = _uniform(seed ^ 0xE2202, n * ln).reshape(n, ln)
    sh = (splitmix64(seed ^ 0x5B1F7, n * ln) % np.uint64(3)).astype(np.uint8).reshape(n, ln) + 1
    return np.where(u < err, (codes + sh) & 3, codes).astype(np.uint8)


def single_end_codes(genome, n_reads, read_len, seed, err=0.005, isnv=()):
    """codes u8[n_reads][read_len] in 0..3."""
    g = CODE[np.frombuffer(bytes(genome), np.uint8)]
    L = len(g)
    r = splitmix64(seed, 2 * n_reads)
    start = (r[0::2] % np.uint64(L - read_len + 1)).astype(np.int64)
    rev = (r[1::2] >> np.uint64(63)).astype(bool)
    codes = g[start[:, None] + np.arange(read_len, dtype=np.int64)[None, :]]
    _apply_isnv(codes, start, read_len, isnv, seed, n_reads)
    codes = _apply_errors(codes, seed, err)
    codes[rev] = (3 - codes[rev])[:, ::-1]
    return np.ascontiguousarray(codes)


def paired_codes(genome, n_pairs, read_len, seed, err=0.005, isnv=(), frag_len=300):
    g = CODE[np.frombuffer(bytes(genome), np.uint8)]
    L = len(g)
    r = splitmix64(seed, 2 * n_pairs)
    start = (r[0::2] % np.uint64(L - frag_len + 1)).astype(np.int64)
    flip = (r[1::2] >> np.uint64(63)).astype(bool)
    frag = g[start[:, None] + np.arange(frag_len, dtype=np.int64)[None, :]]
    _apply_isnv(frag, start, frag_len, isnv, seed, n_pairs)
    frag[flip] = (3 - frag[flip])[:, ::-1]
    r1 = np.ascontiguousarray(frag[:, :read_len])
    r2 = np.ascontiguousarray((3 - frag[:, frag_len - read_len:])[:, ::-1])
    return _apply_errors(r1, seed ^ 0x101, err), _apply_errors(r2, seed ^ 0x202, err)


def codes_to_ascii(codes):
    return [BASES[row].tobytes() for row in codes]


def pack_codes(codes):
    """u8 codes [n][len] -> fixed-stride 2-bit records, the layout of bk_pack_reads: words u32[n][ceil(len/16)]."""
    n, ln = codes.shape
    sw = (ln + 15) // 16
    pad = np.zeros((n, sw * 16), np.uint32)
    pad[:, :ln] = codes
    sh = (2 * np.arange(16, dtype=np.uint32))[None, None, :]
    words = (pad.reshape(n, sw, 16) << sh).sum(axis=2, dtype=np.uint64).astype(np.uint32)
    return words, np.full(n, ln, np.uint16)
