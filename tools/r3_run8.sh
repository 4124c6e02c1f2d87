#!/bin/bash
mkdir -p gpurun_out/r3j
cd "$GRAFT_REPO_ROOT" || exit 1
for fly in 2 3 4 5 6 8; do
  python bench.py --no-cpu-baseline --no-other-configs --in-flight $fly --steps 10 --warmup 4 > gpurun_out/r3j/bench_fly$fly.json 2> gpurun_out/r3j/bench_fly$fly.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3j/bench_fly*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.3g ms/sample %.4f serial %.4f scan solo %.4f fly %.4f" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_ms_in_flight_incl_queueing"]))
    except Exception as e: print(f, "unreadable", e)
PY
