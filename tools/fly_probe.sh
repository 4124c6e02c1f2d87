# samples in flight x hardware queues (config 2 unless BENCH_ARGS says otherwise)
cd $GRAFT_REPO_ROOT
for f in 3 4 5 6 8; do
  for q in 8 16; do
  echo -n "in flight $f queues $q: "
  GPU_MAX_HW_QUEUES=$q timeout 150 python3 bench.py ${BENCH_ARGS:---steps 10 --warmup 4} --no-cpu-baseline --no-other-configs --in-flight $f 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3), d['ms_per_sample'], d['serial_ms_per_sample'])"
  done
done
