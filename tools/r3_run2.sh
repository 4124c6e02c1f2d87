#!/bin/bash
mkdir -p gpurun_out/r3b
cd "$GRAFT_REPO_ROOT" || exit 1
( time python -m pytest tests -m gpu -q --durations=8 ) > gpurun_out/r3b/pytest.log 2>&1
tail -25 gpurun_out/r3b/pytest.log
( time python bench.py ) > gpurun_out/r3b/bench_default.json 2> gpurun_out/r3b/bench_default.err
tail -c 1500 gpurun_out/r3b/bench_default.err
for combo in "2 2" "2 4" "3 6" "1 2" "1 4"; do
  set -- $combo
  python bench.py --no-cpu-baseline --no-other-configs --share $1 --in-flight $2 --steps 8 --warmup 3 > gpurun_out/r3b/bench_c2_share$1_fly$2.json 2> gpurun_out/r3b/bench_c2_share$1_fly$2.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.3g ms/sample %.4f serial %.4f scan solo %.4f fly %.4f" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_ms_in_flight_incl_queueing"]))
        if "value_with_k0" in d: print("  k0:", d["value_with_k0"])
        if "other_configs" in d:
            for k,v in d["other_configs"].items(): print("  ", k, "%.3g" % v["value"], v["ms_per_sample"], v["scan"])
    except Exception as e: print(f, "unreadable", e)
PY
