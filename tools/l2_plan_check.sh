# the planned Level-2 grid (l2_plan_kernel): its parity test, the headline, configs 4 and 3, and away from the sweet spot with four in flight
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "level2_on_as_many or planes_are_clean or batched_pushes or hpv_ragged" 2>&1 | tail -3
for i in 1 2; do timeout 150 python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2', '%.4g' % d['value'], '%.4f' % d['ms_per_sample'], '%.4f' % d['serial_ms_per_sample'], d['check'])"; done
timeout 200 python bench.py --no-cpu-baseline --no-other-configs --config 4 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 4', '%.4g' % d['value'], '%.4f' % d['ms_per_sample'])"
timeout 200 python bench.py --no-cpu-baseline --no-other-configs --config 4 --steps 3 --warmup 1 --in-flight 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 4, four in flight', '%.4g' % d['value'], '%.4f' % d['ms_per_sample'])"
timeout 200 python3 tools/stress_probe.py release "0.5 %,5 %,random" 4 2>&1 | grep " bp"
