#!/bin/bash
# bin_count_kernel under its ablation switches (testing build, BK_BIN_ABLATE: 1 no items read, 2 nothing written, 3 no E atomics,
# 4 no V writes), one config-2 sample at a time under rocprofv3.   gpurun -- bash tools/bin_ablate.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in ${ABS:-0 1 2 3 4}; do
  BK_BIN_ABLATE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pb$ab -- python3 tools/scan_ablate.py 2 0 > /dev/null 2>&1
  echo "BK_BIN_ABLATE=$ab $(grep 'bin_count' $(find gpurun_out/pb$ab -name '*kernel_stats.csv') | cut -d, -f1-4)"
  rm -rf gpurun_out/pb$ab
done
