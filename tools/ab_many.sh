#!/bin/bash
# bench.py (BENCH_ARGS, default config 2) with each of the given builds of libbronko_hip.so in turn, N rounds.   gpurun -- bash tools/ab_many.sh N lib...
cd "$GRAFT_REPO_ROOT" || exit 1
N=$1; shift
for i in $(seq $N); do
  for lib in "$@"; do
    BRONKO_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-other-configs ${BENCH_ARGS:---steps 10 --warmup 4} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.ljust(36), '%.4g' % d['value'], '%.4f' % d['ms_per_sample'], '%.4f' % d['serial_ms_per_sample'], {k: round(x, 4) for k, x in d['kernels_ms_per_sample_solo'].items()})"
  done
done
