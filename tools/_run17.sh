cd $GRAFT_REPO_ROOT
bash tools/round_fuzz.sh > /dev/null 2>&1
tail -8 gpurun_out/r05_fuzz.txt
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/r05_gpu_suite.txt 2>&1
tail -8 gpurun_out/r05_gpu_suite.txt
