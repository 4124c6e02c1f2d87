#!/bin/bash
# scratch: whole GPU suite + fuzz
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1 | tail -10
BK_VERIFY_ANSWERS=1 python tools/fuzz_parity.py 400 2024 2>&1 | tail -2
