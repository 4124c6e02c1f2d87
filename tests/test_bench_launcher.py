"""bench.py's launcher and input generators on CPU: `--gpus N` starts N ranks itself (gloo here), refuses to report an
N-GPU number it cannot measure, refuses to run with testing aids in the environment; the torch generators bench.py
uses on the GPU give the numpy generator's reads bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np

from bronko_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if not k.startswith("BK_") and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=300)


def test_gpus_2_spawns_two_gloo_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--dry-run", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry_run"] is True


def test_gpus_2_without_two_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        return   # a multi-GPU box can run it
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "GPU" in r.stderr


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--dry-run", "--backend", "gloo"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_refuses_testing_aids_in_the_environment():
    r = _run(["--dry-run"], env={"BK_SCAN_ABLATE": "1"})
    assert r.returncode != 0 and "BK_SCAN_ABLATE" in r.stderr


def test_torch_generators_equal_numpy_generators(sars_paths):
    ref = synth.read_fasta_bytes(sars_paths[0])
    g, isnv = synth.sample_genome(ref, 2)
    c = synth.single_end_codes(g, 3000, 150, 77, err=0.01, isnv=isnv)
    t = synth.single_end_codes_torch(g, 3000, 150, 77, err=0.01, isnv=isnv)
    assert np.array_equal(c, t.numpy().astype(np.uint8))
    # any slice of the stream can be generated on its own (bench.py makes config 3's 10 M pairs in batches)
    t2 = synth.single_end_codes_torch(g, 700, 150, 77, err=0.01, isnv=isnv, row0=1300)
    assert np.array_equal(c[1300:2000], t2.numpy().astype(np.uint8))
    w, l = synth.pack_codes(c)
    tw, tl = synth.pack_codes_torch(t)
    assert np.array_equal(w.view(np.int32), tw.numpy()) and np.array_equal(l.view(np.int16), tl.numpy())
    a, b = synth.paired_codes(g, 2000, 150, 5, err=0.01, isnv=isnv)
    ta, tb = synth.paired_codes_torch(g, 2000, 150, 5, err=0.01, isnv=isnv)
    assert np.array_equal(a, ta.numpy().astype(np.uint8)) and np.array_equal(b, tb.numpy().astype(np.uint8))
    ta2, tb2 = synth.paired_codes_torch(g, 300, 150, 5, err=0.01, isnv=isnv, row0=1234)
    assert np.array_equal(a[1234:1534], ta2.numpy().astype(np.uint8)) and np.array_equal(b[1234:1534], tb2.numpy().astype(np.uint8))
