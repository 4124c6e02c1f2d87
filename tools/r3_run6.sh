#!/bin/bash
mkdir -p gpurun_out/r3h
cd "$GRAFT_REPO_ROOT" || exit 1
( time python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/r3h/pytest.log 2>&1
tail -5 gpurun_out/r3h/pytest.log
python tools/scan_ablate.py 2 2>&1 | grep config
bash tools/r3_prof.sh r3h 2>&1 | grep -E "bk::|\[bk\]"
python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r3h/bench_c2.json 2> gpurun_out/r3h/bench_c2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3h/bench_c2.json").read().strip().splitlines()[-1])
print("value %.3g ms/sample %.4f serial %.4f scan solo %.4f fly %.4f" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_ms_in_flight_incl_queueing"]), d["kernels_ms_per_sample_solo"], d["check"])
PY
