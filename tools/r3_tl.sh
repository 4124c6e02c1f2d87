#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 --in-flight 1 --steps 1 --warmup 1 --samples-per-step 2 --no-cpu-baseline > /dev/null 2>&1
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv") > gpurun_out/c5_literal_timeline.txt
rm -rf gpurun_out/tl5
tail -22 gpurun_out/c5_literal_timeline.txt
