cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r2l > gpurun_out/r2l_profile.log 2>&1
tail -14 gpurun_out/r2l_profile.log
cp gpurun_out/r2l_pmc_traffic.json profiles/pmc_traffic.json
bash tools/round_bench_lines.sh r2l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 --selected-only --in-flight 1 --steps 1 --warmup 1 --samples-per-step 4 --no-cpu-baseline > /dev/null 2>&1
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv") > gpurun_out/tl5.txt
rm -rf gpurun_out/tl5
cat gpurun_out/tl5.txt | tail -8
