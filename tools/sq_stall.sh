# Where the waves of the path's kernels spend their cycles (one config-2 sample at a time): issue vs wait, by instruction class.
#   gpurun -- bash tools/sq_stall.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="--steps 1 --warmup 1 --samples-per-step 8 --no-cpu-baseline --no-other-configs --in-flight 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_sq_a -- python3 bench.py $P > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq_b -- python3 bench.py $P > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b -name "*counter_collection.csv") | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items():
    print(k[:48], {a:round(b/1e6,3) for a,b in v.items() if isinstance(b,(int,float)) and a != 'launches'})
"
rm -rf gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b
