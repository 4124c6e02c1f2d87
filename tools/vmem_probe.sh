# memory-instruction counters of every kernel of one config-2 sample at a time (release build): how many vector memory instructions,
# L2 requests.   gpurun -- bash tools/vmem_probe.sh
# (the TA_* and TCP_* counters are not asked for: on this pool a pass with them never ends -- each ran into its timeout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="--steps 1 --warmup 1 --samples-per-step 8 --no-cpu-baseline --no-other-configs --in-flight 1"
i=0
for C in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/vm_$i -- python3 bench.py $P > gpurun_out/vm_$i.log 2>&1 || echo "pass $i failed or timed out: $C"
done
python3 tools/pmc_summary.py $(find gpurun_out/vm_* -name "*counter_collection.csv") > gpurun_out/vmem_probe.json
rm -rf gpurun_out/vm_[0-9] gpurun_out/vm_*.log
python3 - <<PY
import json
d = json.load(open("gpurun_out/vmem_probe.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
    if v["launches"] >= 8: print(" ", k[:40].ljust(40), {a: round(b / 1e6, 3) for a, b in v.items() if a != "launches" and not a.startswith("hbm")})
PY
