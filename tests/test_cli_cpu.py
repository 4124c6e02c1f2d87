"""`bronko build` through the binary (the three invocations of /root/reference/tests/build_tests.rs) and the
argument checks that need no GPU."""
import os
import subprocess

from bronko_amd.hostlib import HostIndex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BRONKO = os.path.join(ROOT, "bronko_amd", "bin", "bronko")


def run(*args):
    return subprocess.run([BRONKO] + list(args), capture_output=True, text=True)


def test_build_tests_rs(golden_dir, sars_paths, tmp_path, oracle):
    out = str(tmp_path / "bronko")
    r = run("build", "-g", *sars_paths, "-t", "2", "-o", out)            # build_tests.rs:8-21
    assert r.returncode == 0 and r.stdout.startswith("bronko v0.1.0")
    a = HostIndex.load(out + ".bkdb")
    assert a.k == 21 and a.n_files == 4 and a.n_entries == 2501142
    r = run("build", "-g", os.path.join(golden_dir, "HPV16.fa"), "-k", "19", "-t", "2", "-o", out)   # :23-35
    assert r.returncode == 0 and HostIndex.load(out + ".bkdb").k == 19
    r = run("build", "-g", os.path.join(golden_dir, "HPV16.fa"), "-t", "2", "-o", out)               # :37-47
    assert r.returncode == 0
    mine, gold = oracle.Index.load(out + ".bkdb"), oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    assert mine.entries().tobytes() == gold.entries().tobytes() and mine.files() == gold.files()
    assert "finished in" in r.stderr                                       # main.rs:28


def test_build_argument_errors(golden_dir, tmp_path):
    hp = os.path.join(golden_dir, "HPV16.fa")
    for args in (["build", "-g", hp, "-k", "22"], ["build", "-g", hp, "-k", "13"], ["build", "-g", hp, "-k", "33"],
                 ["build", "-g", hp, "-t", "0"], ["build", "-g", hp, "-t", "100000"], ["build", "-g", "x.txt"]):
        r = run(*args, "-o", str(tmp_path / "o"))
        assert r.returncode == 1 and "ERROR" in r.stdout, args
    assert run("build").returncode == 2                                     # arg_required_else_help
    assert run("frobnicate").returncode == 2


def test_call_without_gpu_fails_loudly(golden_dir, tmp_path):
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    fq = str(tmp_path / "x.fastq")
    open(fq, "w").write("@a\nACGT\n+\nIIII\n")
    r = run("call", "-d", os.path.join(golden_dir, "hpv.bkdb"), "-r", fq, "-o", str(tmp_path / "o"))
    assert r.returncode == 1 and "no HIP device" in r.stdout
