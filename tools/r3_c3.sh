#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 ) 2>&1 | tail -6
BK_VERIFY_ANSWERS=1 python tools/fuzz_parity.py 500 555 2>&1 | tail -2
python3 tools/l2_stats.py 3 2>&1 | grep -E "bk\] scan marked" | cut -c1-300
python3 bench.py --config 3 --no-cpu-baseline --no-other-configs > gpurun_out/c3.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-other-configs > gpurun_out/c2.json 2>/dev/null
for f in c2 c3; do python3 - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f", "%.4g"%d["value"], "ms/step", round(d["ms_per_step"],3), "serial", d.get("serial_ms_per_sample"), d.get("kernels_ms_per_sample_solo"))
PY
done
