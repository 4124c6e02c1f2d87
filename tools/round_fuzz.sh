#!/bin/bash
# the round's kept fuzz runs (profiles/r05_fuzz.txt; r04_fuzz.txt, r03_fuzz.txt before): three seeds x 1000 iterations of tools/fuzz_parity.py at the given sources
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
  echo "tools/fuzz_parity.py, libbronko_hip_testing.so built from sources $(python -c 'import bench; print(bench.source_build_id())')"
  for seed in ${SEEDS:-41 42 43}; do
    echo "== fuzz_parity.py 1000 $seed"
    python tools/fuzz_parity.py 1000 $seed 2>&1 | tail -4
  done
} > gpurun_out/${OUT:-r05_fuzz}.txt 2>&1
tail -12 gpurun_out/${OUT:-r05_fuzz}.txt
