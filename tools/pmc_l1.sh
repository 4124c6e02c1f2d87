cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BK_SCAN_ABLATE=1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_r1l_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pmc_r1l_sq -name "*counter_collection.csv") | head -12
unset BK_SCAN_ABLATE
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_r1m_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pmc_r1m_sq -name "*counter_collection.csv") | head -12
