import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
