cd $GRAFT_REPO_ROOT
SHORT=1 bash tools/final_captures.sh r06_b > gpurun_out/r06_b_final_captures.log 2>&1
tail -30 gpurun_out/r06_b_final_captures.log
bash tools/config5_pmc.sh r06_b > gpurun_out/r06_b_config5_pmc.log 2>&1
tail -5 gpurun_out/r06_b_config5_pmc.log | cut -c1-300
