cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1k -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_r1k_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r1k_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r1k_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_r1k_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pmc_r1k_fetch gpurun_out/pmc_r1k_write -name "*counter_collection.csv") > gpurun_out/r1k_pmc_hbm.json
python3 tools/pmc_summary.py $(find gpurun_out/pmc_r1k_sq -name "*counter_collection.csv") > gpurun_out/r1k_pmc_sq.json
tail -1 gpurun_out/prof_r1k_bench.log | cut -c1-900
head -c 1500 gpurun_out/r1k_pmc_hbm.json
