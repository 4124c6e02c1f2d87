# Wave-instructions per kernel of two builds (A = bronko_amd/libbronko_hip.so, B = _ab/libbronko_hip.so; tools/ab.sh) on bench.py's
# config 2, one sample in flight.   gpurun -- bash tools/ab_insts.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in A B; do
  if [ $v = B ]; then export BRONKO_HIP_LIB=$PWD/_ab/libbronko_hip.so; else unset BRONKO_HIP_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pab$v -- python3 bench.py --steps 1 --warmup 1 --samples-per-step 8 --no-cpu-baseline --no-other-configs --in-flight 1 > /dev/null 2>&1
  python3 tools/pmc_summary.py $(find gpurun_out/pab$v -name "*counter_collection.csv") | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in sorted(d.items()):
    print('$v', k[:44].ljust(44), {a.replace('SQ_',''):round(b/1e6,3) for a,b in v.items() if isinstance(b,(int,float)) and a != 'launches'})
"
  rm -rf gpurun_out/pab$v
done
