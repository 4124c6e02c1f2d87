#!/bin/bash
# A/B of two builds of libbronko_hip.so (A = the tree's, B = _ab/libbronko_hip.so) on other configs: tools/ab_cfg.sh N "bench args"
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-2}; shift
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export BRONKO_HIP_LIB=$PWD/_ab/libbronko_hip.so; else unset BRONKO_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4g' % d['value'], '%.4f' % d['ms_per_sample'], '%.4f' % d['serial_ms_per_sample'], {k: round(x, 4) for k, x in d['kernels_ms_per_sample_solo'].items()})"
  done
done
