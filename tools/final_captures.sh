# The captures of a round's last build that are kept under profiles/ (run on the GPU box: gpurun -- bash tools/final_captures.sh r04_a).
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards (tools/README.md).
cd $GRAFT_REPO_ROOT
T=${1:-r06_a}
bash tools/profile_round.sh $T > gpurun_out/${T}_profile.log 2>&1
tail -14 gpurun_out/${T}_profile.log
cp gpurun_out/${T}_pmc_traffic.json profiles/pmc_traffic.json
# the plain bench lines: the default run (config 2 + K0 figure + other_configs + CPU baseline) and each config at full length
( time python3 bench.py ) > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
python3 bench.py --config 3 --no-other-configs > gpurun_out/${T}_bench_config3.json 2> /dev/null
python3 bench.py --config 4 --steps 5 --warmup 2 > gpurun_out/${T}_bench_config4.json 2> /dev/null
python3 bench.py --config 5 --steps 3 --warmup 1 > gpurun_out/${T}_bench_config5.json 2> /dev/null
python3 bench.py --config 5 --steps 3 --warmup 1 --selected-only --no-cpu-baseline > gpurun_out/${T}_bench_config5_selected_only.json 2> /dev/null
for f in default config3 config4 config5 config5_selected_only; do tail -1 gpurun_out/${T}_bench_$f.json | cut -c1-220; done
# (SHORT=1: a second capture of a round -- the bench lines, the profiles and the config-5 timelines only)
if [ -z "$SHORT" ]; then
# where the scan's time goes (testing build), what is left to Level 2, ingest
{ python3 tools/scan_ablate.py 2; python3 tools/scan_ablate.py 3 0,10,2,6,7,9,11; } 2>&1 | grep config > gpurun_out/${T}_scan_ablate.txt
{ for c in 2 3 5; do python3 tools/l2_stats.py $c 2>&1 | grep -E "bk\]|config"; done; } > gpurun_out/${T}_l2_stats.txt
python3 tools/ingest_bench.py 2>&1 | tail -3 > gpurun_out/${T}_ingest.txt
fi
# one config-5 sample's kernels in order (selected-only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 --selected-only --in-flight 1 --steps 1 --warmup 1 --samples-per-step 4 --no-cpu-baseline > /dev/null
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv") > gpurun_out/${T}_config5_selected_only_timeline.txt
rm -rf gpurun_out/tl5
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 --in-flight 1 --steps 1 --warmup 1 --samples-per-step 2 --no-cpu-baseline > /dev/null
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv") > gpurun_out/${T}_config5_timeline.txt
rm -rf gpurun_out/tl5
tail -8 gpurun_out/${T}_config5_selected_only_timeline.txt
[ -n "$SHORT" ] && exit 0
cat gpurun_out/${T}_scan_ablate.txt

# round 6: away from the sweet spot, the scale rows, the host side of the ingest, engine creation, the kept fuzz runs
python3 tools/stress_probe.py 2>&1 | grep " bp" > gpurun_out/${T}_stress.txt
{ for n in 100 250 500; do python3 tools/scale_probe.py $n 4 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/${T}_scale.txt
bash tools/pack_probe.sh 2>&1 | tail -10 > gpurun_out/${T}_pack_probe.txt
python3 tools/create_timing.py 100 31 2>&1 | grep -E "bk_engine_create|index build" > gpurun_out/${T}_create_timing.txt
cat gpurun_out/${T}_stress.txt gpurun_out/${T}_scale.txt
