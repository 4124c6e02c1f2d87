# Config 2: the kernels of one sample in order (one sample at a time) and, with three in flight, every kernel's duration and overlap.
#   gpurun -- bash tools/timeline2.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--steps 2 --warmup 1 --samples-per-step 64 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl2 -- python3 bench.py $B --in-flight 1 > /dev/null 2>&1
python3 tools/sample_timeline.py $(find gpurun_out/tl2 -name "*kernel_trace.csv")
rm -rf gpurun_out/tl2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl2 -- python3 bench.py $B > /dev/null 2>&1
python3 tools/trace_overlap.py $(find gpurun_out/tl2 -name "*kernel_trace.csv") 128 32
rm -rf gpurun_out/tl2
