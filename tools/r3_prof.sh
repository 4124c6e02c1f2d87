#!/bin/bash
# kernel trace (one sample at a time) of config 2 + L2 statistics
T=${1:-r3x}
mkdir -p gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof_serial -- python3 bench.py --steps 2 --warmup 1 --samples-per-step 32 --no-cpu-baseline --no-other-configs --in-flight 1 > gpurun_out/$T/serial_bench.log 2>&1
cp $(find gpurun_out/$T/prof_serial -name "*kernel_stats.csv") gpurun_out/$T/serial_kernel_stats.csv
rm -rf gpurun_out/$T/prof_serial
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/$T/serial_kernel_stats.csv")))
for r in rows[:14]:
    print("%-90s calls %6s avg_us %9.2f total%% %s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
python tools/l2_stats.py 2 2>&1 | tail -4
