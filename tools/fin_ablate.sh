# finalize kernels of bk_finalize_lean.hip under their ablation switches (testing build), one config-2 sample at a time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in ${ABS:-0 2 3 4}; do
  BK_NO_LEAN_FINALIZE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf$ab -- python3 tools/scan_ablate.py 2 0 > /dev/null 2>&1
  echo "BK_NO_LEAN_FINALIZE=$ab $(grep -E 'ecell|vbin|finalize_exact_kernel|finalize_variant' $(find gpurun_out/pf$ab -name '*kernel_stats.csv') | cut -d, -f1-4 | tr '\n' ' ')"
  rm -rf gpurun_out/pf$ab
done
