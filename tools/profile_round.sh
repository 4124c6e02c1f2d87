# The rocprofv3 captures behind profiles/r04_<tag>_* (r03_<tag>_*, r02_<tag>_* before): kernel stats of the default bench (3 samples in flight) and of the same
# samples one at a time, a kernel trace for tools/trace_overlap.py, three PMC passes (one sample at a time, so that the counters
# of a kernel are not mixed with a co-running one), and the kernel stats of config 3.  Run on the GPU box:
#   gpurun -- bash tools/profile_round.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r2a}
B="--steps 2 --warmup 1 --samples-per-step 64 --no-cpu-baseline --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -- python3 bench.py $B > gpurun_out/prof_${T}_bench.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_serial -- python3 bench.py $B --in-flight 1 > gpurun_out/prof_${T}_serial_bench.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_c3 -- python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --in-flight 1 > gpurun_out/prof_${T}_c3_bench.log 2>&1
# ... the same steps fed from sequence lines through K0 (pack_words_kernel / pack_slow_kernel), one sample at a time
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_k0 -- python3 bench.py $B --in-flight 1 --from-ascii > gpurun_out/prof_${T}_k0_bench.log 2>&1
P="--steps 1 --warmup 1 --samples-per-step 8 --no-cpu-baseline --no-other-configs --in-flight 1"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_fetch -- python3 bench.py $P > /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_write -- python3 bench.py $P > /dev/null
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_sq -- python3 bench.py $P > /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_k0f -- python3 bench.py $P --from-ascii > /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${T}_k0w -- python3 bench.py $P --from-ascii > /dev/null
python3 tools/pmc_summary.py $(find gpurun_out/pmc_${T}_k0f gpurun_out/pmc_${T}_k0w -name "*counter_collection.csv") > gpurun_out/${T}_k0_pmc_hbm.json
python3 tools/pmc_summary.py $(find gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write -name "*counter_collection.csv") > gpurun_out/${T}_pmc_hbm.json
python3 tools/pmc_summary.py $(find gpurun_out/pmc_${T}_sq -name "*counter_collection.csv") > gpurun_out/${T}_pmc_sq.json
python3 tools/trace_overlap.py $(find gpurun_out/prof_$T -name "*kernel_trace.csv") 128 32 > gpurun_out/${T}_overlap.txt
cp $(find gpurun_out/prof_$T -name "*kernel_stats.csv") gpurun_out/${T}_kernel_stats.csv
cp $(find gpurun_out/prof_${T}_serial -name "*kernel_stats.csv") gpurun_out/${T}_serial_kernel_stats.csv
cp $(find gpurun_out/prof_${T}_c3 -name "*kernel_stats.csv") gpurun_out/${T}_config3_serial_kernel_stats.csv
cp $(find gpurun_out/prof_${T}_k0 -name "*kernel_stats.csv") gpurun_out/${T}_k0_serial_kernel_stats.csv
grep "^{" gpurun_out/prof_${T}_k0_bench.log | tail -1 > gpurun_out/${T}_k0_serial_bench_under_rocprof.json
grep "^{" gpurun_out/prof_${T}_bench.log | tail -1 > gpurun_out/${T}_bench_under_rocprof.json
grep "^{" gpurun_out/prof_${T}_serial_bench.log | tail -1 > gpurun_out/${T}_serial_bench_under_rocprof.json
grep "^{" gpurun_out/prof_${T}_c3_bench.log | tail -1 > gpurun_out/${T}_config3_serial_bench_under_rocprof.json
python3 tools/make_pmc_traffic.py gpurun_out/${T}_pmc_hbm.json gpurun_out/${T}_pmc_sq.json $T > gpurun_out/${T}_pmc_traffic.json
rm -rf gpurun_out/prof_${T}_k0 gpurun_out/pmc_${T}_k0f gpurun_out/pmc_${T}_k0w gpurun_out/prof_$T gpurun_out/prof_${T}_serial gpurun_out/prof_${T}_c3 gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write gpurun_out/pmc_${T}_sq
cut -c1-400 gpurun_out/${T}_bench_under_rocprof.json
cat gpurun_out/${T}_overlap.txt
