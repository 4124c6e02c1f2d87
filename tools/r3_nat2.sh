#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_gpu_parity.py -x -q  2>&1 | tail -3 ) 2>&1 | tail -6
python3 bench.py --config 5 --steps 3 --warmup 1 --selected-only --no-cpu-baseline > gpurun_out/c5s.json 2>/dev/null
python3 - <<PY
import json
d=json.loads(open("gpurun_out/c5s.json").read().strip().splitlines()[-1])
print("c5s", "%.4g"%d["value"], "ms/step", round(d["ms_per_step"],3), d["roofline"].get("avg_kernel_ms"))
print({k:v for k,v in d.items() if "kernel" in k or "serial" in k})
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 --selected-only --in-flight 1 --steps 1 --warmup 1 --samples-per-step 4 --no-cpu-baseline > /dev/null 2>&1
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv") > gpurun_out/c5_timeline.txt
rm -rf gpurun_out/tl5
tail -24 gpurun_out/c5_timeline.txt
