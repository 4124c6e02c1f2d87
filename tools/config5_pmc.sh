# Config 5 (100 strains, k = 31) counter passes, one sample at a time: FETCH / WRITE / SQ / L2 hit / atomics per kernel, literal
# and selected-only; round 6: the same passes for config 3 (four strains, 10 x 1 M pairs).  Every bk:: kernel is kept (tools/pmc_summary.py).
#   gpurun -- bash tools/config5_pmc.sh r06_a   ->  gpurun_out/<tag>_config5_pmc_*.json, <tag>_config3_pmc.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r05_a}
timeout 400 rocprofv3 -L > gpurun_out/${T}_counters_list.txt 2>&1
P="--config 5 --steps 1 --warmup 1 --samples-per-step 3 --no-cpu-baseline --in-flight 1"
for mode in literal selected; do
  X=""; [ $mode = selected ] && X="--selected-only"
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc5_${mode}_$i -- python3 bench.py $P $X > gpurun_out/pmc5_${mode}_$i.log 2>&1
  done
  python3 tools/pmc_summary.py $(find gpurun_out/pmc5_${mode}_* -name "*counter_collection.csv") > gpurun_out/${T}_config5_pmc_${mode}.json
  rm -rf gpurun_out/pmc5_${mode}_[0-9]
done
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc3_$i -- python3 bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --in-flight 1 > gpurun_out/pmc3_$i.log 2>&1
done
python3 tools/pmc_summary.py $(find gpurun_out/pmc3_* -name "*counter_collection.csv") > gpurun_out/${T}_config3_pmc.json
rm -rf gpurun_out/pmc3_[0-9] gpurun_out/pmc3_*.log gpurun_out/pmc5_*.log
python3 - <<PY
import json
for m in ("literal", "selected", "config3"):
    d = json.load(open("gpurun_out/${T}_config5_pmc_%s.json" % m if m != "config3" else "gpurun_out/${T}_config3_pmc.json"))
    print(m)
    for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
        print(" ", k[:44].ljust(44), {a: round(b / 1e6, 3) for a, b in v.items() if a != "launches"}, v["launches"])
PY
