cd $GRAFT_REPO_ROOT
python tools/fuzz_parity.py 1 43 999 2>&1 | tail -2 | cut -c1-250
python tools/fuzz_parity.py 600 43 400 2>&1 | tail -2 | cut -c1-250
python -m pytest tests/test_gpu_many_genomes.py -x -q -m gpu 2>&1 | tail -3
