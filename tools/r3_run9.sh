#!/bin/bash
mkdir -p gpurun_out/r3k
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q -k "ascii or packing or ragged or cli or call or lane or ingest or config2" 2>&1 | tail -4
python bench.py --no-cpu-baseline > gpurun_out/r3k/bench_default.json 2> gpurun_out/r3k/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3k/bench_default.json").read().strip().splitlines()[-1])
print("value %.3g ms/sample %.4f serial %.4f scan solo %.4f fly %.4f" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_ms_in_flight_incl_queueing"]), d["kernels_ms_per_sample_solo"])
print("  k0:", d["value_with_k0"])
for k,v in d["other_configs"].items(): print("  ", k, "%.3g" % v["value"], "%.3f" % v["ms_per_sample"], "%.3f" % v["serial_ms_per_sample"], v["scan"]["avg_kernel_ms"], v["kernels_ms_per_sample_solo"])
PY
python tools/ingest_bench.py 2>&1 | tail -4
