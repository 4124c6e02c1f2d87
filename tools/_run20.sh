cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_cli.py tests/test_pargz.py -x -q 2>&1 | tail -4
{ python3 tools/cli_end_to_end.py 32 1000000 64 1 2>&1 | grep -v "^\[" | tail -5; python3 tools/cli_end_to_end.py 1 10000000 64 1 2>&1 | grep -v "^\[" | tail -4; python3 tools/cli_end_to_end.py 64 1000000 64 100 2>&1 | grep -v "^\[" | tail -5; } > gpurun_out/r05_cli_end_to_end.txt 2>&1
cat gpurun_out/r05_cli_end_to_end.txt
nproc; uptime
