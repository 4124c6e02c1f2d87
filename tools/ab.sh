#!/bin/bash
# A/B of two builds of libbronko_hip.so in one GPU run: A = bronko_amd/libbronko_hip.so, B = _ab/libbronko_hip.so (a copy made
# before the sources were changed, or built from a variant).  Alternates them N times on bench.py's config 2.
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-3}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export BRONKO_HIP_LIB=$PWD/_ab/libbronko_hip.so; else unset BRONKO_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4g' % d['value'], '%.4f' % d['ms_per_sample'], '%.4f' % d['serial_ms_per_sample'], {k: round(x, 4) for k, x in d['kernels_ms_per_sample_solo'].items()})"
  done
done
