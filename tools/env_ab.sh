#!/bin/bash
# bench.py (BENCH_ARGS, default config 2) on the testing build with and without one BK_* switch, N rounds.
#   gpurun -- bash tools/env_ab.sh 2 BK_NO_FINALIZE_SIDE=1
cd "$GRAFT_REPO_ROOT" || exit 1
N=$1; shift
export BRONKO_HIP_LIB=$PWD/bronko_amd/libbronko_hip_testing.so
for i in $(seq $N); do
  for v in "" "$@"; do
    env $v python bench.py --no-cpu-baseline --no-other-configs ${BENCH_ARGS:---steps 10 --warmup 4} --experiment 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-(none)}'.ljust(28), '%.4g' % d['value'], '%.4f' % d['ms_per_sample'], '%.4f' % d['serial_ms_per_sample'], {k: round(x, 4) for k, x in d['kernels_ms_per_sample_solo'].items()})"
  done
done
