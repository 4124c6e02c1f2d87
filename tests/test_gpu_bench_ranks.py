"""bench.py's N > 1 paths through the real engines on the one GPU of the test box: `--gpus 2 --backend gloo` starts two ranks that
share the device (rank r uses GPU r mod #GPUs; on a multi-GPU node the same lines run over RCCL), each with its own engine:
config 2 (a sample = one shard per rank, reduce-scatter of the packed counter planes, sharded finalize, combine), config 4 (one
sample's batches dealt to the ranks) and config 5 (whole samples per rank, no collective).  What the line reports as `check` -- the
last sample's perfect / variant k-mers and scanned k-mer occurrences -- must be what ONE rank reports for the same reads, and what
the oracle computes from the reads both ranks generated (no multi-GPU node was available to any round: this is the evidence
that the N > 1 lines compute the right thing before their speed can be measured)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from bronko_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
READS = 20000


def _bench(args):
    e = {k: v for k, v in os.environ.items() if not k.startswith("BK_") and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, BENCH, "--reads", str(READS), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"] + args,
                       capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def _oracle_check(oracle, sars_paths, mates):
    ix = oracle.Index.build(21, [sars_paths[0]])
    try:
        pile = oracle.sample_pileup(ix, mates)
        return [int(x) for x in pile.stats.sum(axis=0)[:, 0]], [int(x) for x in pile.stats.sum(axis=0)[:, 1]]
    finally:
        ix.close()


def test_config2_two_ranks_share_each_sample(oracle, sars_paths):
    """Weak scaling: a sample is one 20,000-read shard per rank.  The two-rank line's check = the oracle on both shards' reads."""
    two = _bench(["--gpus", "2", "--backend", "gloo", "--batches", "1", "--samples-per-step", "2"])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["scaling"] == "weak"
    assert two["check"]["kmers_scanned"] == [2 * READS * 130]
    assert two["comm"]["width_bits"] in (16, 32, 64)
    genome, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[0]), 2)
    reads = []
    for rank in range(2):   # bench.py: batch b of rank r has seed 2 * 1000003 + r + 7919 b
        reads += synth.codes_to_ascii(synth.single_end_codes(genome, READS, 150, 2 * 1000003 + rank, err=0.005, isnv=isnv))
    perfect, variant = _oracle_check(oracle, sars_paths, [reads])
    assert two["check"]["perfect_kmers"] == perfect and two["check"]["variant_kmers"] == variant
    one = _bench(["--batches", "1", "--samples-per-step", "2"])
    assert one["n_gpus"] == 1 and one["check"]["kmers_scanned"] == [READS * 130]


def test_config4_batches_dealt_to_two_ranks(oracle, sars_paths):
    """Strong scaling: ONE sample of four batches; one rank pushes all four, two ranks two each -- same counts."""
    one = _bench(["--config", "4", "--batches", "4"])
    two = _bench(["--config", "4", "--batches", "4", "--gpus", "2", "--backend", "gloo"])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["scaling"] == "strong"
    assert one["check"] == two["check"] and one["check"]["kmers_scanned"] == [4 * READS * 130]
    genome, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[0]), 4)
    reads = []
    for b in range(4):      # bench.py: batch b has seed 4 * 1000003 + 7919 b
        reads += synth.codes_to_ascii(synth.single_end_codes(genome, READS, 150, 4 * 1000003 + 7919 * b, err=0.005, isnv=isnv))
    perfect, variant = _oracle_check(oracle, sars_paths, [reads])
    assert one["check"]["perfect_kmers"] == perfect and one["check"]["variant_kmers"] == variant


def test_config5_whole_samples_per_rank():
    """Many samples against many strains: whole samples per rank, no collective; rank 0's last sample is sample 0 either way."""
    args = ["--config", "5", "--strains", "12", "--samples-per-step", "1"]
    one = _bench(args + ["--batches", "1"])
    two = _bench(args + ["--batches", "2", "--gpus", "2", "--backend", "gloo"])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2
    assert one["check"] == two["check"] and one["check"]["kmers_scanned"] == [READS * 120]
    assert max(one["check"]["perfect_kmers"]) > 1000
    assert "no collective" in two["config"]["parallelism"]
