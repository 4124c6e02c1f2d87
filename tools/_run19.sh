cd $GRAFT_REPO_ROOT
SEEDS="44 45 46 47 48 49 50 51" OUT=r05_fuzz_more bash tools/round_fuzz.sh > /dev/null 2>&1
tail -20 gpurun_out/r05_fuzz_more.txt
