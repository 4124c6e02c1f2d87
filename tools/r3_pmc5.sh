#!/bin/bash
# scratch: SQ counters of config 5's kernels (selected-only), one sample at a time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmc_sq_x -- python3 bench.py --config 5 --steps 1 --warmup 1 --samples-per-step 4 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pmc_sq_x -name "*counter_collection.csv") | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict) and ('finalize' in k or 'scan' in k or 'level2' in k): print(k[:45], {a:round(b/1e6,3) for a,b in v.items() if isinstance(b,(int,float))})
"
rm -rf gpurun_out/pmc_sq_x
