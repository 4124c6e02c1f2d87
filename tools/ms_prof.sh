cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 20 --no-cpu-baseline 2>&1 | tail -1 | grep -o "\"value\": [0-9.]*\|kernels_ms[^}]*}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ms -- python3 tools/many_strains_check.py 30 200000 --no-oracle > gpurun_out/prof_ms.log 2>&1
tail -3 gpurun_out/prof_ms.log
find gpurun_out/prof_ms -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-120
