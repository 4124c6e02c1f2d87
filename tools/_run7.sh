cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "many_strains or selected_genome or file_bitmaps or planes_are_clean or k31 or multi_sequence or reverse_complement or config5" 2>&1 | tail -5
python tools/fuzz_parity.py 300 71 2>&1 | tail -3
python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
python3 bench.py --config 5 --steps 2 --warmup 1 --selected-only --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for X in "" "--selected-only"; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 $X --in-flight 1 --steps 1 --warmup 1 --samples-per-step 3 --no-cpu-baseline > /dev/null 2>&1
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv")
rm -rf gpurun_out/tl5
done
