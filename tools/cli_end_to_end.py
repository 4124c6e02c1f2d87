#!/usr/bin/env python3
"""`bronko call` end to end: FASTQ.gz files on disk -> VCFs, wall time of the whole command (gunzip + parse on host threads,
PCIe, the GPU path, calls, output files).  Writes S synthetic samples of N reads (config-2 shape: wuhan_ref, 150 bp single-end,
0.5 % errors) as real .fastq.gz files under a scratch directory, then times the binary with 1 lane and one inflate
thread per file (zlib's gzread), 1 lane and the inflate threads -t allows (pargz.hpp), the same with the files read ahead of their
turn (the binary's default), and the default lanes.
With a fourth argument N > 1 the references are N synthetic strains (wuhan_ref + 300 substitutions each, k = 31: BASELINE
config 5's shape) written as FASTA files, sample s is derived from strain s mod N.
usage: tools/cli_end_to_end.py [samples 16] [reads 1000000] [threads 32] [strains 1]"""
import gzip, os, subprocess, sys, tempfile, time
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bronko_amd import synth

REF = os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
BIN = os.path.join(ROOT, "bronko_amd", "bin", "bronko")


def write_sample(arg):
    path, n, seed, base = arg
    g, isnv = synth.sample_genome(base, seed)
    codes = synth.single_end_codes(g, n, 150, 7000003 + seed, isnv=isnv)
    seqs = synth.BASES[codes]                                   # u8 [n][150]
    rec = np.empty((n, 4 + 8 + 1 + 150 + 3 + 150 + 1), np.uint8)   # "@r" + 8 hex digits + \n seq \n+\n qual \n
    hexd = np.frombuffer(b"0123456789abcdef", np.uint8)
    idx = np.arange(n, dtype=np.uint32)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2] = ord("e"); rec[:, 3] = ord("a")
    for d in range(8):
        rec[:, 4 + d] = hexd[(idx >> np.uint32(4 * (7 - d))) & np.uint32(15)]
    rec[:, 12] = 10
    rec[:, 13:163] = seqs
    rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
    q = np.random.default_rng(seed).random((n, 150))            # binned qualities as current instruments write them
    rec[:, 166:316] = np.where(q < 0.90, ord("F"), np.where(q < 0.96, ord(":"), np.where(q < 0.99, ord(","), ord("#"))))
    rec[:, 316] = 10
    with gzip.open(path, "wb", compresslevel=1) as f:
        f.write(rec.tobytes())
    return os.path.getsize(path)


def prepare(tmp, S, N, NS):
    """Writes the reference FASTA files (NS > 1: synthetic strains) and S samples of N reads as .fastq.gz under tmp.
    Returns (reference paths, extra arguments, sample paths)."""
    wuhan = synth.read_fasta_bytes(REF)
    if NS > 1:
        files = synth.strain_files(wuhan, NS)
        refs, bases, kk = [], [], ["-k", "31"]
        for name, seqs in files:
            fp = os.path.join(tmp, name + ".fasta")
            with open(fp, "wb") as f:
                for sn, sq in seqs:
                    f.write(b">" + sn.encode() + b"\n" + bytes(sq) + b"\n")
            refs.append(fp); bases.append(bytes(seqs[0][1]))
    else:
        refs, bases, kk = [REF], [wuhan], []
    paths = [os.path.join(tmp, "sample%02d.fastq.gz" % s) for s in range(S)]
    t0 = time.time()
    with Pool(min(S, 16)) as pool:
        sizes = pool.map(write_sample, [(p, N, 100 + s, bases[s % len(bases)]) for s, p in enumerate(paths)])
    print("wrote %d samples x %d reads: %.1f MB of .fastq.gz (%.1f MB of FASTQ text) in %.0f s" % (S, N, sum(sizes) / 1e6, S * N * 317 / 1e6, time.time() - t0), flush=True)
    return refs, kk, paths


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    NS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    tmp = tempfile.mkdtemp(prefix="bronko_e2e_")
    refs, kk, paths = prepare(tmp, S, N, NS)
    # (lanes per device, inflate threads per file, files read ahead of their turn): None = the binary's own choice
    runs = [("1", "1", False), ("1", None, False), ("1", None, True), (None, None, True)] if S > 1 else [("1", "1", False), ("1", None, False), ("1", None, True)]
    if os.environ.get("E2E_EXTRA"):   # more (lanes, inflate threads) pairs with the files read ahead: E2E_EXTRA="2:2,2:4,4:2"
        runs = [("1", "1", False)] + [(a.split(":")[0] or None, a.split(":")[1] or None, True) for a in os.environ["E2E_EXTRA"].split(",")]
    for lanes, inflate, ahead in runs:
        env = dict(os.environ)
        env.pop("BRONKO_LANES", None); env.pop("BRONKO_INFLATE_THREADS", None); env.pop("BRONKO_NO_READ_AHEAD", None)
        if lanes:
            env["BRONKO_LANES"] = lanes
        if inflate:
            env["BRONKO_INFLATE_THREADS"] = inflate
        if not ahead:
            env["BRONKO_NO_READ_AHEAD"] = "1"
        out = os.path.join(tmp, "out_%s_%s_%d" % (lanes or "default", inflate or "default", ahead))
        t0 = time.time()
        r = subprocess.run([BIN, "call", "-g"] + refs + ["-r"] + paths + kk + ["-t", str(T), "-o", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t0
        if r.returncode != 0:
            print(r.stderr.decode()[-2000:])
            raise SystemExit("bronko call failed")
        n_vcf = len([f for f in os.listdir(out) if f.endswith(".vcf")])
        how = [ln.split("] ", 1)[-1] for ln in r.stdout.decode().splitlines() if "inflated on" in ln]
        print("bronko call, %d reference genome(s), %s lanes per device, %s, files %s (-t %d): %.2f s wall for %d samples (%d VCFs) = %.2f M reads/s end to end" %
              (len(refs), lanes or "default", how[0] if how else "one inflate thread per file", "read ahead" if ahead else "read in their turn", T, dt, S, n_vcf, S * N / dt / 1e6), flush=True)
    bodies = [open(os.path.join(tmp, "out_%s_%s_%d" % (l or "default", i or "default", ah), "sample00.vcf")).read().split("\n", 3)[-1] for l, i, ah in runs]
    print("same VCF body in every run:", all(b == bodies[0] for b in bodies))
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
