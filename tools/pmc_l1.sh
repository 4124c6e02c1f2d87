# SQ instruction counters of the scan kernel under each ablation (BK_SCAN_ABLATE): 1 = Level 1 alone, 4 = without Level 2, 0 = complete
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in 1 4 0; do
  export BK_SCAN_ABLATE=$ab
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_ab$ab -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== BK_SCAN_ABLATE=$ab"
  python3 tools/pmc_summary.py $(find gpurun_out/pmc_ab$ab -name "*counter_collection.csv") | grep -i -E "kernel|scan" | head -4
done
