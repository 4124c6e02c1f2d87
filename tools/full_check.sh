# the whole GPU suite and the round's three fuzz runs at the current sources:  gpurun -- bash tools/full_check.sh
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/last_tests.txt
cat gpurun_out/last_tests.txt
OUT=r06_fuzz_c timeout 2400 bash tools/round_fuzz.sh
