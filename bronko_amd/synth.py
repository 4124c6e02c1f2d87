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


# ---- the same streams with torch tensors (any device): bench.py and the full-size tests generate their inputs on the GPU ----
# int64 tensors hold the u64 values bit for bit (+, *, ^ wrap identically); logical shifts and unsigned remainders are spelled out.

def _t_shr(z, s):
    import torch  # noqa: F401
    return (z >> s) & ((1 << (64 - s)) - 1)


def _t_splitmix64(seed, n, device, start=0):
    """outputs start+1 .. start+n of the splitmix64 stream started at `seed`, as int64 bit patterns"""
    import torch
    def s64(v):
        v &= _M
        return v - (1 << 64) if v >= (1 << 63) else v
    z = torch.arange(start + 1, start + n + 1, dtype=torch.int64, device=device) * s64(0x9E3779B97F4A7C15) + s64(seed)
    z = (z ^ _t_shr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ _t_shr(z, 27)) * s64(0x94D049BB133111EB)
    return z ^ _t_shr(z, 31)


def _t_umod(z, m):
    """z (u64 bit pattern in int64) mod m, 0 < m < 2^31"""
    hi, lo = _t_shr(z, 32), z & 0xffffffff
    return (hi * ((1 << 32) % m) + lo) % m


def _t_uniform(seed, n, device, start=0):
    import torch
    return _t_shr(_t_splitmix64(seed, n, device, start), 11).to(torch.float64) * (1.0 / (1 << 53))


def _t_apply_isnv(codes, start, span, isnv, seed, n, row0):
    import torch
    for j, (p, alt, af) in enumerate(isnv):
        hit = (start <= p) & (p < start + span)
        rows = torch.nonzero(hit).flatten()
        if rows.numel() == 0:
            continue
        u = _t_uniform(seed ^ (0x15A70000 + j), n, codes.device, row0)[rows]
        rows = rows[u < af]
        codes[rows, p - start[rows]] = alt


def _t_apply_errors(codes, seed, err, row0):
    import torch
    if err <= 0:
        return codes
    n, ln = codes.shape
    u = _t_uniform(seed ^ 0xE2202, n * ln, codes.device, row0 * ln).reshape(n, ln)
    sh = _t_umod(_t_splitmix64(seed ^ 0x5B1F7, n * ln, codes.device, row0 * ln), 3).reshape(n, ln) + 1
    return torch.where(u < err, (codes + sh) & 3, codes)


def _t_genome(genome, device):
    import torch
    return torch.from_numpy(CODE[np.frombuffer(bytes(genome), np.uint8)].astype(np.int64)).to(device)


def single_end_codes_torch(genome, n_reads, read_len, seed, err=0.005, isnv=(), device="cpu", row0=0):
    """rows [row0, row0 + n_reads) of single_end_codes(genome, N, ...) for any N >= row0 + n_reads, as an int64 tensor"""
    import torch
    g = _t_genome(genome, device)
    L = g.numel()
    r = _t_splitmix64(seed, 2 * n_reads, device, 2 * row0)
    start = _t_umod(r[0::2], L - read_len + 1)
    rev = _t_shr(r[1::2], 63) != 0
    codes = g[start[:, None] + torch.arange(read_len, dtype=torch.int64, device=device)[None, :]]
    _t_apply_isnv(codes, start, read_len, isnv, seed, n_reads, row0)
    codes = _t_apply_errors(codes, seed, err, row0)
    codes[rev] = (3 - codes[rev]).flip(1)
    return codes


def paired_codes_torch(genome, n_pairs, read_len, seed, err=0.005, isnv=(), frag_len=300, device="cpu", row0=0):
    import torch
    g = _t_genome(genome, device)
    L = g.numel()
    r = _t_splitmix64(seed, 2 * n_pairs, device, 2 * row0)
    start = _t_umod(r[0::2], L - frag_len + 1)
    flip = _t_shr(r[1::2], 63) != 0
    frag = g[start[:, None] + torch.arange(frag_len, dtype=torch.int64, device=device)[None, :]]
    _t_apply_isnv(frag, start, frag_len, isnv, seed, n_pairs, row0)
    frag[flip] = (3 - frag[flip]).flip(1)
    r1 = frag[:, :read_len].contiguous()
    r2 = (3 - frag[:, frag_len - read_len:]).flip(1).contiguous()
    return _t_apply_errors(r1, seed ^ 0x101, err, row0), _t_apply_errors(r2, seed ^ 0x202, err, row0)


def pack_codes_torch(codes):
    """int64 codes [n][len] -> (words int32 [n][ceil(len/16)], lens int16 [n]): the record layout of bk_pack_reads"""
    import torch
    n, ln = codes.shape
    sw = (ln + 15) // 16
    pad = torch.zeros((n, sw * 16), dtype=torch.int64, device=codes.device)
    pad[:, :ln] = codes
    sh = (2 * torch.arange(16, dtype=torch.int64, device=codes.device))[None, None, :]
    w = (pad.reshape(n, sw, 16) << sh).sum(dim=2)
    w = torch.where(w >= (1 << 31), w - (1 << 32), w).to(torch.int32)
    return w.contiguous(), torch.full((n,), ln, dtype=torch.int16, device=codes.device)


def strain_files(base, n_strains, n_sub=300, seed0=5000):
    """BASELINE config 5's index: n_strains synthetic strains = `base` + n_sub seeded substitutions each (about 1 % of a
    SARS-CoV-2 genome; strain 0 included), as the metadata list HostIndex.build_mem / the oracle's Index.build_mem take."""
    files = []
    for s in range(n_strains):
        g = np.frombuffer(bytes(base), np.uint8).copy()
        r = splitmix64(seed0 + s, 2 * n_sub)
        pos = (r[0::2] % np.uint64(len(g))).astype(np.int64)
        sh = (r[1::2] % np.uint64(3)).astype(np.int64) + 1
        for p, d in zip(pos, sh):
            g[p] = BASES[(int(CODE[g[p]]) + int(d)) & 3]
        files.append(("strain%03d" % s, [("seq%03d" % s, g.tobytes())]))
    return files
