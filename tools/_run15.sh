cd $GRAFT_REPO_ROOT
bash tools/final_captures.sh r05_c > gpurun_out/r05_c_final.log 2>&1
tail -30 gpurun_out/r05_c_final.log
bash tools/config5_pmc.sh r05_c > gpurun_out/r05_c_config5_pmc.log 2>&1
python3 tools/stress_probe.py > gpurun_out/r05_stress.txt 2>&1
SEEDS="41 42 43" OUT=r05_fuzz bash tools/round_fuzz.sh
