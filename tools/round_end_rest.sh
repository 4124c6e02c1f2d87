# the rest of final_captures.sh at the round's last sources (tag r06_c), every step under its own timeout
cd $GRAFT_REPO_ROOT
T=r06_c
{ timeout 200 python3 tools/scan_ablate.py 2; timeout 200 python3 tools/scan_ablate.py 3 0,10,2,6,7,9,11; } 2>&1 | grep config > gpurun_out/${T}_scan_ablate.txt
{ for c in 2 3 5; do timeout 200 python3 tools/l2_stats.py $c 2>&1 | grep -E "bk\]|config"; done; } > gpurun_out/${T}_l2_stats.txt
timeout 200 python3 tools/ingest_bench.py 2>&1 | tail -3 > gpurun_out/${T}_ingest.txt
timeout 300 bash tools/pack_probe.sh 2>&1 | tail -10 > gpurun_out/${T}_pack_probe.txt
timeout 200 python3 tools/create_timing.py 100 31 2>&1 | grep -E "bk_engine_create|index build" > gpurun_out/${T}_create_timing.txt
wc -l gpurun_out/${T}_scan_ablate.txt gpurun_out/${T}_l2_stats.txt gpurun_out/${T}_ingest.txt gpurun_out/${T}_pack_probe.txt gpurun_out/${T}_create_timing.txt
cat gpurun_out/${T}_scan_ablate.txt gpurun_out/${T}_ingest.txt | cut -c1-200
tail -1 gpurun_out/${T}_create_timing.txt
