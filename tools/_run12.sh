cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_calls.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -4
