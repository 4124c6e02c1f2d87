cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BRONKO_HIP_LIB=$PWD/bronko_amd/libbronko_hip_testing.so
B="--experiment --steps 2 --warmup 1 --samples-per-step 64 --no-cpu-baseline --no-other-configs --in-flight 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py $B > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/ks/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bk::" in r["Name"]: print(r["Name"][:60].ljust(60), r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
T=$(find gpurun_out/ks -name "*kernel_trace.csv"); python3 tools/sample_timeline.py $T zero_small_kernel 100; python3 tools/sample_timeline.py $T zero_small_kernel 101
rm -rf gpurun_out/ks
