import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # built artefacts are not tracked: a fresh checkout builds them (make is incremental, a no-op when they are current)
    built = [os.path.join(ROOT, "bronko_amd", f) for f in ("libbronko_hip.so", "libbronko_hip_testing.so", "libbronko_host.so",
                                                           os.path.join("bin", "bronko"))] + [os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(b) for b in built):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_first():
    """Some GPU tests view engine buffers as torch tensors.  torch must initialise its HIP context before the engine library
    has spun up its own streams and threads in the process (initialising it afterwards has failed with "No HIP GPUs are
    available" depending on test order), so do it once at session start; a no-op on a box without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    except Exception:   # noqa: BLE001 -- CPU-only runs do not care
        pass
    yield


@pytest.fixture
def testing_lib():
    """Engines created inside the test bind libbronko_hip_testing.so, the -DBK_TESTING build of the same sources: the BK_*
    environment variables that force a code path (a small LDS window, launch splitting, ...) exist only there."""
    from bronko_amd import _ffi
    _ffi.use_testing_library(True)
    yield
    _ffi.use_testing_library(False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


SARS_ORDER = ["wuhan_ref.fasta", "OM223929.1.fasta", "ON765678.1.fasta", "PX392231.1.fasta"]  # tests/build_tests.rs:11-14


@pytest.fixture(scope="session")
def sars_paths():
    return [os.path.join(GOLDEN, "4_sarscov2", n) for n in SARS_ORDER]
