cd $GRAFT_REPO_ROOT
cat /proc/loadavg; python tools/create_timing.py 100 31 2>&1 | grep -v "gathered votes" | tail -45
timeout 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_parity.py -x -q -m gpu -k "build or golden or many_strains or k31 or k_variants or sarscov2 or hpv_single or config5" 2>&1 | tail -4
python tools/fuzz_parity.py 150 91 2>&1 | tail -2
python3 bench.py --config 5 --steps 1 --warmup 1 --selected-only --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['setup_s'])"
