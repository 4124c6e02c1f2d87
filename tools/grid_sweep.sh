#!/bin/bash
# bench.py's config 2 (samples in flight) with the binned scan on fewer workgroups than CUs (testing build, BK_ITEM_GRID).
#   gpurun -- bash tools/grid_sweep.sh "256 224 192" "3 4"
cd "$GRAFT_REPO_ROOT" || exit 1
export BRONKO_HIP_LIB=$PWD/bronko_amd/libbronko_hip_testing.so
for f in ${2:-3}; do
for g in ${1:-256 248 240 224 208 192 160}; do
  BK_ITEM_GRID=$g python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 4 --in-flight $f --experiment 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight $f grid $g', '%.4g' % d['value'], '%.4f' % d['ms_per_sample'], '%.4f' % d['serial_ms_per_sample'], {k: round(x, 4) for k, x in d['kernels_ms_per_sample_solo'].items()})"
done
done
