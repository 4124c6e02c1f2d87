#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 tools/l2_stats.py 5 2>&1 | grep -E "own files|finalize:" | head
