#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 ) 2>&1 | tail -6
BK_VERIFY_ANSWERS=1 python tools/fuzz_parity.py 500 31337 2>&1 | tail -2
bash tools/r3_tl.sh
