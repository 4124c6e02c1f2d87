# (every pass under its own timeout: a set of TCP_* counters once hung a whole call)
# counters of config 5's every-genome mode only (a quick look between two builds; tools/config5_pmc.sh is the full capture)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="--config 5 --steps 1 --warmup 1 --samples-per-step 3 --no-cpu-baseline --in-flight 1"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc5_p_$i -- python3 bench.py $P > gpurun_out/pmc5_p_$i.log 2>&1
done
python3 tools/pmc_summary.py $(find gpurun_out/pmc5_p_* -name "*counter_collection.csv") > gpurun_out/c5_pmc_probe.json
rm -rf gpurun_out/pmc5_p_[0-9] gpurun_out/pmc5_p_*.log
python3 - <<PY
import json
d = json.load(open("gpurun_out/c5_pmc_probe.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0))[:8]:
    print(" ", k[:44].ljust(44), {a: round(b / 1e6, 3) for a, b in v.items() if a != "launches"}, v["launches"])
PY
