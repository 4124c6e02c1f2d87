#!/bin/bash
mkdir -p gpurun_out/r3i
cd "$GRAFT_REPO_ROOT" || exit 1
( time python bench.py ) > gpurun_out/r3i/bench_default.json 2> gpurun_out/r3i/bench_default.err
tail -c 300 gpurun_out/r3i/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3i/bench_default.json").read().strip().splitlines()[-1])
print("value %.3g ms/sample %.4f serial %.4f scan solo %.4f fly %.4f" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_ms_in_flight_incl_queueing"]), d["kernels_ms_per_sample_solo"])
print("  k0:", d["value_with_k0"]["value"], d["value_with_k0"]["ms_per_sample"])
for k,v in d["other_configs"].items(): print("  ", k, "%.3g" % v["value"], "%.3f" % v["ms_per_sample"], "%.3f" % v["serial_ms_per_sample"], v["scan"]["avg_kernel_ms"], v["kernels_ms_per_sample_solo"])
PY
python tools/scan_ablate.py 3 0,2,6,7,9 2>&1 | grep config
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py 2>&1 | tail -3
