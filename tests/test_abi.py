"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/bronko_hip.h
declares, the host-side packer (K0) follows the KMC read-splitting contract, and the engine refuses to
run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from bronko_amd import _ffi, pack_reads, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bronko_hip.h")).read()
    declared = set(re.findall(r"\b(bk_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_ffi.SYMBOLS)
    for testing in (False, True):   # the release library and its -DBK_TESTING twin
        L = _ffi.load(testing=testing)
        for s in declared:
            assert hasattr(L, s), s
        assert L.bk_abi_version() == 8


def test_integration_md_declares_every_symbol():
    """The Rust `extern "C"` block a bronko maintainer would add names exactly the functions the header declares."""
    hdr = open(os.path.join(ROOT, "include", "bronko_hip.h")).read()
    declared = set(re.findall(r"\b(bk_[a-z_0-9]+)\s*\(", hdr))
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    rust = set(re.findall(r"pub fn (bk_[a-z_0-9]+)\s*\(", block))
    assert rust == declared, (sorted(declared - rust), sorted(rust - declared))


def test_release_library_reads_no_environment_variable():
    """The BK_* testing / measurement aids are compiled into libbronko_hip_testing.so only."""
    rel = open(_ffi.LIB_PATH, "rb").read()
    tst = open(_ffi.TESTING_LIB_PATH, "rb").read()
    for name in (b"BK_SCAN_ABLATE", b"BK_LDS_BINS", b"BK_REF_IN_LDS", b"BK_WINDOW_FILE", b"BK_MAX_LAUNCH_RECORDS", b"BK_L2_COUNT", b"BK_NO_FUSE", b"BK_NO_GATHER",
                 b"BK_NOISE_SERIAL", b"BK_ITEM_CAPS", b"BK_GATHER_ABLATE", b"BK_NO_VOTE_TABLE", b"BK_NO_ISO23", b"BK_SYNC_DEBUG", b"BK_ITEM_SHARE", b"BK_NO_L2_PLAN"):
        assert name not in rel, name
        assert name in tst, name


def test_bucket_info_layout_matches_repr_c():
    assert C.sizeof(_ffi.BucketInfo) == 12                        # build.rs:52-60
    assert _ffi.BucketInfo.location.offset == 4 and _ffi.BucketInfo.idx.offset == 8


def test_call_record_layout():
    assert C.sizeof(_ffi.CallRecord) == 72 and _ffi.CallRecord.pos.offset == 8 and _ffi.CallRecord.af.offset == 56
    assert C.sizeof(_ffi.CallParams) == 72 and _ffi.CallParams.min_af.offset == 16
    assert C.sizeof(_ffi.CallSummary) == 56


def test_pack_reads_contract():
    w, l = pack_reads([b"ACGTACGTAC", b"ACGTNACGTA", b"acgtacg", b"AC", b""], 5)
    assert l.tolist() == [10, 5, 7]                               # runs shorter than k are dropped
    assert w[0, 0] == sum(c << (2 * i) for i, c in enumerate([0, 1, 2, 3, 0, 1, 2, 3, 0, 1]))
    # long runs are cut into chunks overlapping by k-1 so that every k-mer is kept exactly once
    w, l = pack_reads([b"A" * 100], 21, stride_words=2)
    assert l.tolist() == [32] * 6 + [28]
    assert sum(int(x) - 20 for x in l) == 100 - 20


def test_pack_reads_equals_numpy_packing():
    g = synth.read_fasta_bytes(os.path.join(ROOT, "tests", "golden", "HPV16.fa"))
    codes = synth.single_end_codes(g, 500, 150, 9)
    w1, l1 = synth.pack_codes(codes)
    w2, l2 = pack_reads(synth.codes_to_ascii(codes), 21)
    assert np.array_equal(w1, w2) and np.array_equal(l1, l2)


def test_engine_fails_loudly_without_gpu(oracle, golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bronko_amd import BronkoError
    from tests import helpers
    ix = oracle.Index.load(os.path.join(golden_dir, "hpv.bkdb"))
    with pytest.raises(BronkoError) as ei:
        helpers.engine_from_oracle_index(ix)
    assert ei.value.status == -2                                   # BK_ERR_NO_DEVICE
    ix.close()
