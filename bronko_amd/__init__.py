"""bronko_amd -- MI355X (gfx950) k-mer -> pileup engine behind bronko's `call` hot path.

The product is the C-ABI library `libbronko_hip.so` (include/bronko_hip.h) plus the C++ host code under
bronko_amd/host/.  This Python package is a thin ctypes mirror of that ABI used by tests and bench.py; it
has no compute of its own and fails loudly if the HIP library is missing.
"""
from .engine import BronkoError, Engine, Params, build_index_device, lib_path, pack_reads  # noqa: F401
