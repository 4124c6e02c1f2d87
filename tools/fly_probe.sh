cd $GRAFT_REPO_ROOT
for f in 4 6 8; do
  for q in 8 16; do
  echo "in flight $f queues $q"
  GPU_MAX_HW_QUEUES=$q timeout 300 python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --in-flight $f 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_sample'], d['serial_ms_per_sample'])"
  done
done
