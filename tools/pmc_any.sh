# Any PMC counters for the kernels of one config-2 sample at a time (testing build; tools/scan_ablate.py, switch 0).
#   gpurun -- bash tools/pmc_any.sh "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" [kernel substring] [config] [BK_SCAN_ABLATE switch]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d gpurun_out/pany -- python3 tools/scan_ablate.py ${3:-2} ${4:-0} > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pany -name "*counter_collection.csv") | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in sorted(d.items()):
    if '${2:-scan_}' in k: print(k[:40].ljust(40), {a:round(b/1e6,3) for a,b in v.items() if isinstance(b,(int,float)) and a != 'launches'})
"
rm -rf gpurun_out/pany
