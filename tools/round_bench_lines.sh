# The plain bench lines kept under profiles/r02_bench_*.json (run on the GPU box: gpurun -- bash tools/round_bench_lines.sh <tag>)
T=${1:-r2}
python3 bench.py > gpurun_out/${T}_bench_config2.json 2> gpurun_out/${T}_bench_config2.err
python3 bench.py --config 3 > gpurun_out/${T}_bench_config3.json 2> /dev/null
python3 bench.py --config 5 --steps 3 --warmup 1 > gpurun_out/${T}_bench_config5.json 2> /dev/null
python3 bench.py --config 5 --steps 3 --warmup 1 --selected-only --no-cpu-baseline > gpurun_out/${T}_bench_config5_selected_only.json 2> /dev/null
for f in config2 config3 config5 config5_selected_only; do tail -1 gpurun_out/${T}_bench_$f.json | cut -c1-260; done
