"""build_indexes on the device (bk_build_index) against the oracle's host restatement of build.rs:145-231,
which is pinned to the upstream golden index: the same bucket ids, the same offsets, the same BucketInfos in the same order
inside every bucket (bytes and all) -- for the golden HPV16 index, four strains, k = 31 (bucket ids wrap), several sequences per
file, sequences shorter than k and non-ACGT symbols."""
import os

import numpy as np
import pytest

from bronko_amd import build_index_device, synth
from tests import helpers

pytestmark = pytest.mark.gpu


def _same(oracle, k, files):
    ix = oracle.Index.build_mem(k, files)
    ids, off, ent = build_index_device(k, files)
    assert np.array_equal(ids, ix.bucket_ids())
    assert np.array_equal(off, ix.bucket_off())
    assert ent.tobytes() == ix.entries().tobytes()
    ix.close()
    return len(ids)


def test_golden_hpv16_index(oracle, golden_dir):
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    gold = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    ids, off, ent = build_index_device(21, [("HPV16", [("HPV16", g)])])
    assert np.array_equal(ids, gold.bucket_ids()) and np.array_equal(off, gold.bucket_off())
    assert ent.tobytes() == gold.entries().tobytes()
    assert len(ids) == 165603
    gold.close()


def test_four_strains_and_k31(oracle, sars_paths):
    files = [(os.path.basename(p)[:-6], [(os.path.basename(p), synth.read_fasta_bytes(p))]) for p in sars_paths]
    assert _same(oracle, 21, files) > 600000
    assert _same(oracle, 31, files[:2]) > 0
    assert _same(oracle, 15, files[1:3]) > 0


def test_odd_inputs(oracle, golden_dir):
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    files = [("a", [("s0", g[:3000]), ("tiny", b"ACGTACG"), ("s2", g[3000:5000].lower())]),
             ("b", [("n", g[100:400] + b"NNNNRYK" + g[400:900])]),
             ("empty", [])]
    assert _same(oracle, 21, files) > 0
    # nothing to index at all
    ids, off, ent = build_index_device(21, [("x", [("short", b"ACGT")])])
    assert len(ids) == 0 and off.tolist() == [0] and len(ent) == 0


def test_engine_on_a_device_built_index(oracle, golden_dir):
    """An engine created from the device-built arrays maps a sample like one created from the oracle's."""
    from bronko_amd import Engine
    g = synth.read_fasta_bytes(os.path.join(golden_dir, "HPV16.fa"))
    files = [("HPV16", [("HPV16", g)])]
    ids, off, ent = build_index_device(21, files)
    eng = Engine(21, ids, off, ent, files)
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    reads = helpers.hpv_reads(5000, seed=3)
    helpers.assert_same_pileup(helpers.hip_sample(eng, [reads], 21), oracle.sample_pileup(ix, [reads]))
    eng.close()
    ix.close()


def test_device_memory_query():
    """bk_device_memory: free <= total, and an MI355X reports far more than the tables of the tests."""
    from bronko_amd.engine import device_memory
    free, total = device_memory(0)
    assert 0 < free <= total and total > (64 << 30)
