# r06_c: what the planned Level-2 grid changed (one genome file, four in flight): the profiles of config 2, the default bench line, config 4
cd $GRAFT_REPO_ROOT
T=r06_c
bash tools/profile_round.sh $T > gpurun_out/${T}_profile.log 2>&1
tail -12 gpurun_out/${T}_profile.log
cp gpurun_out/${T}_pmc_traffic.json profiles/pmc_traffic.json
( time python3 bench.py ) > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
python3 bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_config4.json 2> /dev/null
for f in default config4; do tail -1 gpurun_out/${T}_bench_$f.json | cut -c1-220; done
timeout 200 python3 tools/stress_probe.py release all 4 2>&1 | grep " bp" > gpurun_out/${T}_stress_four_in_flight.txt
timeout 200 python3 tools/stress_probe.py release 2>&1 | grep " bp" > gpurun_out/${T}_stress.txt
cat gpurun_out/${T}_stress_four_in_flight.txt gpurun_out/${T}_stress.txt
