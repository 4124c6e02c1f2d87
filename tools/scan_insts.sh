# Wave-instructions of the scan kernel under its ablation switches (testing build; tools/scan_ablate.py): what each stage issues.
#   gpurun -- bash tools/scan_insts.sh [config] [switches]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=${1:-2}
for ab in ${2:-0 14 6 7 9}; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pi$ab -- python3 tools/scan_ablate.py $C $ab > /dev/null 2>&1
  python3 tools/pmc_summary.py $(find gpurun_out/pi$ab -name "*counter_collection.csv") | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items():
    if 'scan_' in k: print('ablate $ab', k[:30], {a.replace('SQ_',''):round(b/1e6,2) for a,b in v.items() if isinstance(b,(int,float)) and a != 'launches'})
"
  rm -rf gpurun_out/pi$ab
done
