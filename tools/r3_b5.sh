#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c5.json 2>/dev/null
python3 bench.py --config 5 --steps 3 --warmup 1 --selected-only --no-cpu-baseline > gpurun_out/c5s.json 2>/dev/null
for f in c5 c5s; do python3 - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f", "%.4g"%d["value"], "ms/step", round(d["ms_per_step"],3), "serial", d.get("serial_ms_per_sample"), d.get("kernels_ms_per_sample_solo"), d.get("build"))
PY
done
