"""bronko_amd/host/fastq_pack.hpp (a FASTQ file parsed and 2-bit packed on several threads: what `bronko call` feeds an engine
from; the reference hands its files to KMC with -t threads, /root/reference/src/call.rs:1166-1181): the records are those of the line
loop it replaces -- every fourth line from the second, split at non-ACGT symbols, runs shorter than k dropped -- for gzip and plain
text, any piece size the inflate delivers, CRLF line ends, a last line without its line end, reads with N, empty lines, streams."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PCAT = os.path.join(ROOT, "bronko_amd", "bin", "pack_cat")


@pytest.fixture(scope="module")
def pcat():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bronko_amd", "host"), "../bin/pack_cat"])
    return PCAT


def fastq(n_reads, seed, crlf=False, n_rate=0.0, var_len=False, last_newline=True):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    g = acgt[rng.integers(0, 4, 30000)]
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n_reads):
        ln = int(rng.integers(0, 260)) if var_len else 150
        a = int(rng.integers(0, len(g) - 300))
        s = g[a:a + ln].copy()
        if n_rate and ln:
            m = rng.random(ln) < n_rate
            s[m] = ord("N")
        if i % 97 == 5 and ln:
            s = np.frombuffer(s.tobytes().lower(), dtype=np.uint8)   # (lower case packs like upper case)
        q = bytes([64 if j else 33 + (i % 40) for j in range(ln)]) if ln else b""   # quality lines that begin with '@' now and then
        if i % 5 == 0 and ln:
            q = b"@" + q[1:]
        out.append(b"@r%d extra\n".replace(b"\n", nl) % i + s.tobytes() + nl + b"+" + nl + q + nl)
    t = b"".join(out)
    if not last_newline:
        t = t[:-len(nl)]
    return t


def expected_records(text, k):
    """the line loop's rule, in Python"""
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    recs, reads = [], 0
    for i, ln in enumerate(lines):
        if i % 4 != 1:
            continue
        reads += 1
        ln = ln.rstrip(b"\r")
        run = []
        for c in ln + b"\0":
            if chr(c) in "ACGTacgt":
                run.append(chr(c).upper())
            else:
                if len(run) >= k:
                    recs.append("".join(run))
                run = []
    return recs, reads


def run(pcat, path, k, threads):
    r = subprocess.run([pcat, path, str(k), str(threads)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.decode().split("\n")
    assert lines[-1] == "" and lines[-2].startswith("reads ")
    _, reads, _, records = lines[-2].split()
    assert int(records) == len(lines) - 2
    return lines[:-2], int(reads)


@pytest.mark.parametrize("shape", ["plain", "gzip", "crlf", "ragged with N", "no last line end", "many members"])
def test_parallel_reader_gives_the_line_loops_records(pcat, tmp_path, shape):
    k = 21
    if shape == "ragged with N":
        t = fastq(30000, 3, n_rate=0.01, var_len=True)
    elif shape == "crlf":
        t = fastq(20000, 4, crlf=True)
    elif shape == "no last line end":
        t = fastq(20000, 5, last_newline=False)
    else:
        t = fastq(60000, 6)
    want, n_reads = expected_records(t, k)
    p = str(tmp_path / ("r.fastq.gz" if shape not in ("plain", "no last line end") else "r.fastq"))
    if p.endswith(".gz"):
        if shape == "many members":
            cut = [0, len(t) // 3 + 7, 2 * len(t) // 3 + 11, len(t)]
            data = b"".join(gzip.compress(t[a:b], 6) for a, b in zip(cut[:-1], cut[1:]))
        else:
            data = gzip.compress(t, 6)
        open(p, "wb").write(data)
    else:
        open(p, "wb").write(t)
    for threads in (1, 2, 5, 8):
        got, reads = run(pcat, p, k, threads)
        assert reads == n_reads, (shape, threads)
        assert got == want, (shape, threads, len(got), len(want))


def test_small_slices_and_lines_longer_than_a_piece(pcat, tmp_path):
    """A plain file is taken in 4 MB slices: reads of 40 kb (ten to a slice, most of them across a slice's end) and a run of empty
    lines; k = 31."""
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i in range(400):
        ln = 40000 if i % 7 else 0
        s = acgt[rng.integers(0, 4, ln)].tobytes()
        out.append(b"@long%d\n" % i + s + b"\n+\n" + b"I" * ln + b"\n")
    t = b"".join(out)
    p = str(tmp_path / "long.fastq")
    open(p, "wb").write(t)
    want, n_reads = expected_records(t, 31)
    # (records of reads longer than 65535 bases are cut into chunks; these fit)
    for threads in (1, 4):
        got, reads = run(pcat, p, 31, threads)
        assert reads == n_reads and got == want


def test_a_stream_takes_the_line_loop(pcat, tmp_path):
    t = fastq(5000, 11)
    want, n_reads = expected_records(t, 21)
    gz = str(tmp_path / "s.fastq.gz")
    open(gz, "wb").write(gzip.compress(t))
    fifo = str(tmp_path / "fifo")
    os.mkfifo(fifo)
    writer = subprocess.Popen(["sh", "-c", 'cat "$0" > "$1"', gz, fifo])
    got, reads = run(pcat, fifo, 21, 4)
    assert writer.wait(timeout=60) == 0
    assert reads == n_reads and got == want


def test_a_damaged_file_is_an_error(pcat, tmp_path):
    t = fastq(40000, 12)
    d = bytearray(gzip.compress(t))
    d[len(d) // 2] ^= 0x55
    p = str(tmp_path / "bad.fastq.gz")
    open(p, "wb").write(bytes(d))
    for threads in (1, 4):
        r = subprocess.run([pcat, p, "21", str(threads)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1 and r.stderr


def test_the_readers_packer_is_the_c_abis_packer(pcat, tmp_path):
    """fastq_pack.hpp packs with its own table-driven packer (the short way for a line that is one run); the records must be what
    bk_pack_reads (include/bronko_hip.h, the C ABI's packer: parity-tested against the oracle) makes of the same lines -- ragged
    reads with N and other symbols, lower case, reads shorter than k, at k = 15, 21 and 31."""
    from bronko_amd import pack_reads
    t = fastq(8000, 21, n_rate=0.02, var_len=True)
    lines = t.split(b"\n")
    seqs = [ln for i, ln in enumerate(lines[:-1]) if i % 4 == 1]
    seqs[7] = seqs[7].replace(b"A", b"a").replace(b"G", b"g")
    seqs[11] = seqs[11][:40] + b"-*R" + seqs[11][43:]
    t2 = b"".join(b"@x\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for s in seqs)
    p = str(tmp_path / "odd.fastq")
    open(p, "wb").write(t2)
    for k in (15, 21, 31):
        words, lens = pack_reads(seqs, k)
        want = []
        for r in range(len(lens)):
            want.append("".join("ACGT"[(int(words[r, i >> 4]) >> (2 * (i & 15))) & 3] for i in range(int(lens[r]))))
        got, reads = run(pcat, p, k, 3)
        assert reads == len(seqs)
        assert got == want, (k, len(got), len(want))
