#!/bin/bash
mkdir -p gpurun_out/r3e
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/debug_kvar.py 31 > gpurun_out/r3e/kvar.log 2>&1

tail -40 gpurun_out/r3e/kvar.log; tail -12 gpurun_out/r3e/kvar_abl5.log
