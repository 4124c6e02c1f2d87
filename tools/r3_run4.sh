#!/bin/bash
mkdir -p gpurun_out/r3d
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/debug_kvar.py > gpurun_out/r3d/kvar.log 2>&1
tail -30 gpurun_out/r3d/kvar.log
python tools/l2_stats.py 2 > gpurun_out/r3d/l2stats.log 2>&1
tail -8 gpurun_out/r3d/l2stats.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3d/prof" -o c2 -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-other-configs --in-flight 1 --steps 3 --warmup 2 > "$GRAFT_REPO_ROOT/gpurun_out/r3d/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r3d/prof_bench.err"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/r3d/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {} | cut -c1-200'
