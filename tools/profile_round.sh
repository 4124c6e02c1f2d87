# The rocprofv3 captures behind profiles/r01_h_*: kernel stats of the default bench (3 samples in flight) and of the same steps
# one sample at a time, a kernel trace for tools/trace_overlap.py, and three PMC passes (one sample at a time, so that the
# counters of a kernel are not mixed with a co-running one).  Run on the GPU box: gpurun -- bash tools/profile_round.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r1h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${T}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_serial -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --in-flight 1 > gpurun_out/prof_${T}_serial_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write -name "*counter_collection.csv") > gpurun_out/${T}_pmc_hbm.json
python3 tools/pmc_summary.py $(find gpurun_out/pmc_${T}_sq -name "*counter_collection.csv") > gpurun_out/${T}_pmc_sq.json
python3 tools/trace_overlap.py $(find gpurun_out/prof_$T -name "*kernel_trace.csv") > gpurun_out/${T}_overlap.txt
cp $(find gpurun_out/prof_$T -name "*kernel_stats.csv") gpurun_out/${T}_kernel_stats.csv
cp $(find gpurun_out/prof_${T}_serial -name "*kernel_stats.csv") gpurun_out/${T}_serial_kernel_stats.csv
python3 bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench.json 2> /dev/null
tail -1 gpurun_out/prof_${T}_bench.log | cut -c1-600
cat gpurun_out/${T}_overlap.txt
