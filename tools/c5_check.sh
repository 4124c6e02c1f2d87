python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or selected or strains" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 --selected-only --in-flight 1 --steps 1 --warmup 1 --samples-per-step 4 --no-cpu-baseline > /dev/null 2>&1
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv") | tee gpurun_out/tl5.txt
rm -rf gpurun_out/tl5
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --selected-only | tail -1 | cut -c1-300
