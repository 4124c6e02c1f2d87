#!/bin/bash
# scratch: whole GPU suite + fuzz + bench lines
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) 2>&1 | tail -8
BK_VERIFY_ANSWERS=1 python tools/fuzz_parity.py 400 777 2>&1 | tail -2
python3 bench.py --config 5 --steps 3 --warmup 1 --selected-only --no-cpu-baseline > gpurun_out/c5s.json 2>/dev/null
python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c5.json 2>/dev/null
python3 bench.py --config 3 --no-cpu-baseline --no-other-configs > gpurun_out/c3.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-other-configs > gpurun_out/c2.json 2>/dev/null
for f in c2 c3 c5 c5s; do python3 - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f", "%.4g"%d["value"], "ms/step", round(d["ms_per_step"],3), "serial", d.get("serial_ms_per_sample"), d.get("kernels_ms_per_sample_solo"))
PY
done
