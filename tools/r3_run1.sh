#!/bin/bash
# round 3, first GPU contact: the GPU suite, then config 2 with and without CU-partitioned scans, config 4, the default line
mkdir -p gpurun_out/r3a
cd "$GRAFT_REPO_ROOT" || exit 1
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r3a/pytest.log 2>&1
tail -30 gpurun_out/r3a/pytest.log
for share in 1 3; do
  python bench.py --no-cpu-baseline --no-other-configs --share $share > gpurun_out/r3a/bench_c2_share$share.json 2> gpurun_out/r3a/bench_c2_share$share.err
  tail -c 600 gpurun_out/r3a/bench_c2_share$share.err
done
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench_c4.json 2> gpurun_out/r3a/bench_c4.err
tail -c 600 gpurun_out/r3a/bench_c4.err
( time python bench.py ) > gpurun_out/r3a/bench_default.json 2> gpurun_out/r3a/bench_default.err
tail -c 600 gpurun_out/r3a/bench_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3a/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.3g ms/sample %.4f serial %.4f scan solo %.4f fly %.4f" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_ms_in_flight_incl_queueing"]))
        if "value_with_k0" in d: print("  k0:", d["value_with_k0"])
        if "other_configs" in d:
            for k,v in d["other_configs"].items(): print("  ", k, "%.3g" % v["value"], v["ms_per_sample"], v["scan"])
    except Exception as e: print(f, "unreadable", e)
PY
