#!/bin/bash
mkdir -p gpurun_out/r3c
cd "$GRAFT_REPO_ROOT" || exit 1
( time python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=5 ) > gpurun_out/r3c/pytest.log 2>&1
tail -25 gpurun_out/r3c/pytest.log
python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r3c/bench_c2.json 2> gpurun_out/r3c/bench_c2.err
tail -c 800 gpurun_out/r3c/bench_c2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3c/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.3g ms/sample %.4f serial %.4f scan solo %.4f fly %.4f" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_ms_in_flight_incl_queueing"]), d["kernels_ms_per_sample_solo"], d["check"])
    except Exception as e: print(f, "unreadable", e)
PY
