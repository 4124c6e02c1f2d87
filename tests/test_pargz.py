"""bronko_amd/host/pargz.hpp (one gzip file inflated on several threads -- how `bronko call` reads a sample when it has threads to
spare; the reference hands its FASTQ files to KMC with -t threads, /root/reference/src/call.rs:1166-1181): the same bytes as
Python's gzip module / zlib for every kind of member and block there is, and an error where gzread has one."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAT = os.path.join(ROOT, "bronko_amd", "bin", "pargz_cat")


@pytest.fixture(scope="module")
def cat():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bronko_amd", "host"), "../bin/pargz_cat"])
    return CAT


def fastq_text(n_reads, seed, read_len=150):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, 20000)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i in range(n_reads):
        a = int(rng.integers(0, len(g) - read_len))
        s = acgt[g[a:a + read_len]].copy()
        e = rng.random(read_len) < 0.01
        s[e] = acgt[rng.integers(0, 4, int(e.sum()))]
        q = (rng.integers(0, 8, read_len) * 5 + 35).astype(np.uint8)
        out.append(b"@read%d/1 lane:%d\n" % (i, i % 7) + s.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return b"".join(out)


def run(cat, path, threads, chunk=0):
    r = subprocess.run([cat, path, str(threads)] + ([str(chunk)] if chunk else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout, r.stderr.decode()


def bgzf(data, level=6):
    """What bgzip writes: members of <= 64 KB with their size in an extra field, and an empty last member"""
    out = []
    for i in range(0, len(data), 65280):
        blk = data[i:i + 65280]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(blk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(c) + 25) + c + struct.pack("<II", zlib.crc32(blk), len(blk)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


TEXT = None


def text():
    global TEXT
    if TEXT is None:
        TEXT = fastq_text(40000, 7)   # 13 MB of FASTQ
    return TEXT


@pytest.mark.parametrize("level", [1, 4, 6, 9])
def test_one_member_every_level_and_thread_count(cat, tmp_path, level):
    t = text()
    p = str(tmp_path / "a.fastq.gz")
    open(p, "wb").write(gzip.compress(t, level))
    for threads, chunk in [(1, 0), (2, 0), (8, 0), (8, 70000), (5, 333333), (16, 20000)]:
        rc, out, err = run(cat, p, threads, chunk)
        assert rc == 0, err
        assert out == t, (level, threads, chunk)


def test_members_one_after_the_other_and_bytes_behind_the_last(cat, tmp_path):
    t = text()
    cuts = [0, 1, 300000, 300001, 4000000, 9000000, len(t)]
    data = b"".join(gzip.compress(t[a:b], 6) for a, b in zip(cuts[:-1], cuts[1:]))
    p = str(tmp_path / "m.fastq.gz")
    open(p, "wb").write(data + b"\0" * 77)   # (gzread ignores what is behind the last member)
    for threads, chunk in [(1, 0), (8, 0), (8, 50000), (3, 1 << 20)]:
        rc, out, err = run(cat, p, threads, chunk)
        assert rc == 0, err
        assert out == t
    # a member with a name, a comment and a header CRC
    hdr = b"\x1f\x8b\x08\x1a\0\0\0\0\0\x03" + b"name.fastq\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(t) + co.flush()
    open(p, "wb").write(hdr + body + struct.pack("<II", zlib.crc32(t), len(t) & 0xffffffff))
    assert gzip.decompress(open(p, "rb").read()) == t
    rc, out, err = run(cat, p, 8, 100000)
    assert rc == 0 and out == t, err


def test_stored_fixed_and_empty(cat, tmp_path):
    t = text()[:3000000]
    p = str(tmp_path / "x.gz")
    cases = [gzip.compress(t, 0),                       # stored blocks only: nothing to enter at, one thread decodes
             gzip.compress(b"@r\nACGT\n+\nIIII\n", 9),   # a fixed-code block
             gzip.compress(b"", 9), b""]                  # an empty text, an empty file
    wants = [t, b"@r\nACGT\n+\nIIII\n", b"", b""]
    # stored, dynamic and fixed blocks mixed in one stream (Z_FULL_FLUSH puts an empty stored block between them)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    mixed = b""
    for i in range(0, len(t), 250000):
        mixed += co.compress(t[i:i + 250000]) + co.flush(zlib.Z_FULL_FLUSH if (i // 250000) % 2 else zlib.Z_SYNC_FLUSH)
    mixed += co.flush()
    cases.append(mixed); wants.append(t)
    for data, want in zip(cases, wants):
        open(p, "wb").write(data)
        for threads, chunk in [(1, 0), (8, 0), (8, 40000)]:
            rc, out, err = run(cat, p, threads, chunk)
            assert rc == 0, err
            assert out == want


def test_text_that_is_no_text_and_text_that_repeats(cat, tmp_path):
    rng = np.random.default_rng(3)
    p = str(tmp_path / "b.gz")
    # bytes of every value: no block passes the test of a boundary, the stream decodes from its start
    noisy = rng.integers(0, 256, 3000000, dtype=np.uint8).tobytes() + bytes(2000000)
    # long runs and copies from exactly 32 KB back
    unit = fastq_text(100, 5)[:32768]
    rep = unit * 200 + b"A" * 1000000 + unit * 50
    for t in (noisy, rep):
        open(p, "wb").write(gzip.compress(t, 6))
        for threads, chunk in [(1, 0), (8, 0), (8, 30000)]:
            rc, out, err = run(cat, p, threads, chunk)
            assert rc == 0, err
            assert out == t


def test_bgzf_members(cat, tmp_path):
    t = text()
    p = str(tmp_path / "b.fastq.gz")
    open(p, "wb").write(bgzf(t))
    assert gzip.decompress(open(p, "rb").read()) == t
    for threads in (1, 8):
        rc, out, err = run(cat, p, threads)
        assert rc == 0, err
        assert out == t
    # BGZF members first, an ordinary member behind them
    open(p, "wb").write(bgzf(t[:5000000])[:-28] + gzip.compress(t[5000000:], 6))
    rc, out, err = run(cat, p, 8, 200000)
    assert rc == 0 and out == t, err


def test_damage_is_an_error(cat, tmp_path):
    t = text()
    good = gzip.compress(t, 6)
    p = str(tmp_path / "d.fastq.gz")
    d = bytearray(good); d[len(d) // 2] ^= 0x55                      # a flipped byte in the data
    open(p, "wb").write(bytes(d))
    with pytest.raises(Exception):
        gzip.decompress(bytes(d))
    for threads in (1, 8):
        rc, out, err = run(cat, p, threads, 100000)
        assert rc == 1 and "damaged" in err
        assert out[:4000000] == t[:4000000]                          # (as with gzread, what decodes is delivered until the error shows)
    open(p, "wb").write(good[:-4000])                                # cut short
    for threads in (1, 8):
        rc, out, err = run(cat, p, threads, 100000)
        assert rc == 1 and t.startswith(out), err
    d = bytearray(good); d[-6] ^= 1                                  # the CRC in the trailer
    open(p, "wb").write(bytes(d))
    rc, out, err = run(cat, p, 8)
    assert rc == 1 and "CRC" in err
    open(p, "wb").write(b"plain text, no gzip\n")
    rc, out, err = run(cat, p, 8)
    assert rc == 1


def _fixed_block(items):
    """One final fixed-Huffman deflate block (RFC 1951 3.2.6) of literals (int) and copies ((3, distance), distance 1 or 97..128)"""
    bits = []
    def huff(code, n):                                               # Huffman codes go in most significant bit first,
        bits.extend((code >> (n - 1 - i)) & 1 for i in range(n))
    def plain(v, n):                                                 # everything else least significant bit first
        bits.extend((v >> i) & 1 for i in range(n))
    plain(1, 1); plain(1, 2)
    for it in items:
        if isinstance(it, int):
            assert it < 144
            huff(0x30 + it, 8)
        else:
            ln, dist = it
            assert ln == 3
            huff(257 - 256, 7)
            if dist == 1:
                huff(0, 5)
            else:
                assert 97 <= dist <= 128
                huff(13, 5); plain(dist - 97, 5)
    huff(0, 7)
    bits += [0] * (-len(bits) % 8)
    return bytes(sum(b << i for i, b in enumerate(bits[j:j + 8])) for j in range(0, len(bits), 8))


def test_a_distance_does_not_reach_into_the_member_before(cat, tmp_path):
    """zlib starts every member with an empty window: a copy from before the member's first byte is "invalid distance too far back",
    however much text the file had before it"""
    t = text()
    a, b = t[:3000000], t[3000000:6000000]
    head = b"\x1f\x8b\x08\0\0\0\0\0\0\xff"
    def member(items, out):
        return head + _fixed_block(items) + struct.pack("<II", zlib.crc32(out), len(out))
    good = member([65, (3, 1)], b"AAAA")
    assert gzip.decompress(good) == b"AAAA"
    p = str(tmp_path / "m.fastq.gz")
    open(p, "wb").write(gzip.compress(a, 6) + good + gzip.compress(b, 6))
    for threads, chunk in [(1, 0), (8, 0), (8, 100000), (3, 250000)]:
        rc, out, err = run(cat, p, threads, chunk)
        assert rc == 0 and out == a + b"AAAA" + b, (threads, chunk, err)
    # ... from 1 and from 100 bytes back, as a member's first thing and its second; the trailer is the one a reader that carried
    # the window over would find right
    for items, lenient in (([(3, 1)], a[-1:] * 3), ([(3, 100)], a[-100:-97]), ([65, (3, 100)], b"A" + a[-99:-96])):
        bad = member(items, lenient)
        with pytest.raises(Exception):
            gzip.decompress(bad)
        with pytest.raises(Exception):
            gzip.decompress(gzip.compress(a[-1000:]) + bad)
        for tail in (b"", gzip.compress(b, 6)):
            open(p, "wb").write(gzip.compress(a, 6) + bad + tail)
            for threads, chunk in [(1, 0), (8, 0), (8, 100000), (3, 250000)]:
                rc, out, err = run(cat, p, threads, chunk)
                assert rc == 1 and "damaged" in err, (items, len(tail), threads, chunk, rc, err)
                assert a.startswith(out[:len(a)]) and len(out) <= len(a) + 1


def test_random_texts_members_and_chunkings(cat, tmp_path):
    """Forty random files: FASTQ-like, FASTA-like and mixed text of random sizes, compression levels, member cuts and flush points,
    read back with random thread counts and chunk sizes -- always the bytes zlib gives."""
    rng = np.random.default_rng(20261003)
    p = str(tmp_path / "r.gz")
    base = fastq_text(12000, 3)
    for it in range(40):
        kind = int(rng.integers(0, 4))
        n = int(rng.integers(1, 3500000))
        if kind == 0:
            t = base[:n]
        elif kind == 1:                                              # FASTA: long lines of ACGT with few repeats
            t = b">seq\n" + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)) + b"\n"
        elif kind == 2:                                              # reads that repeat each other closely (amplicon depth)
            unit = base[:int(rng.integers(500, 40000))]
            t = (unit * (n // len(unit) + 1))[:n]
        else:                                                        # text with stretches that are none
            t = base[:n // 2] + bytes(rng.integers(0, 256, int(rng.integers(1, 70000)), dtype=np.uint8)) + base[n // 2:n]
        level = int(rng.choice([1, 2, 5, 6, 9]))
        cuts = sorted(set([0, len(t)] + [int(x) for x in rng.integers(0, len(t) + 1, int(rng.integers(0, 4)))]))
        data = b""
        for a_, b_ in zip(cuts[:-1], cuts[1:]):
            co = zlib.compressobj(level, zlib.DEFLATED, 31)
            part = t[a_:b_]
            fl = sorted(set(int(x) for x in rng.integers(0, len(part) + 1, int(rng.integers(0, 3)))))
            prev = 0
            for f_ in fl:
                data += co.compress(part[prev:f_]) + co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
                prev = f_
            data += co.compress(part[prev:]) + co.flush()
        if not data:
            data = gzip.compress(b"")
        open(p, "wb").write(data)
        assert gzip.decompress(data) == t
        threads = int(rng.choice([1, 2, 3, 8, 13]))
        chunk = int(rng.choice([0, 0, 17000, 65536, 250000, 1 << 20]))
        rc, out, err = run(cat, p, threads, chunk)
        assert rc == 0, (it, err)
        assert out == t, (it, kind, n, level, cuts, threads, chunk)


def test_streams_are_read_in_one_pass_without_a_probe(cat, tmp_path):
    """`bronko call -r <(zcat x.gz)` / a FIFO / /dev/fd/N: an input that is not a regular file can be read once and has no size.  The
    line reader (fastx.hpp GzLineReader, what the binary opens its inputs with) must not probe it for the gzip magic (the probe's
    three bytes are gone from the stream) nor map it (0 lines, rc 0): it stays with zlib's gzread, gzip or plain text,
    whatever the thread count; a regular file of the same content gives the same lines."""
    t = fastq_text(4000, 5)
    lines = t
    gz = str(tmp_path / "s.fastq.gz")
    plain = str(tmp_path / "s.fastq")
    open(gz, "wb").write(gzip.compress(t))
    open(plain, "wb").write(t)
    for src in (gz, plain):
        for threads in (1, 4):
            r = subprocess.run([cat, "--lines", src, str(threads)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0 and r.stdout == lines, (src, threads, r.stderr)
            fifo = str(tmp_path / "fifo")
            if os.path.exists(fifo):
                os.unlink(fifo)
            os.mkfifo(fifo)
            writer = subprocess.Popen(["sh", "-c", 'cat "$0" > "$1"', src, fifo])
            r = subprocess.run([cat, "--lines", fifo, str(threads)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
            assert writer.wait(timeout=60) == 0
            assert r.returncode == 0 and r.stdout == lines, (src, threads, "fifo", len(r.stdout), r.stderr)
    # the mapped reader itself refuses what it cannot map, loudly
    fifo = str(tmp_path / "fifo2")
    os.mkfifo(fifo)
    writer = subprocess.Popen(["sh", "-c", 'cat "$0" > "$1"', gz, fifo])
    rc, out, err = run(cat, fifo, 4)
    writer.kill()
    writer.wait()
    assert rc == 1 and "regular file" in err
