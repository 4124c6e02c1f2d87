cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_calls.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pn -- python3 -m pytest tests/test_gpu_calls.py -q -m gpu -k "hpv_single or four_strains" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pn/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "noise" in r["Name"] or "call_kernel" in r["Name"]: print(r["Name"][:50].ljust(50), r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3), r["MaxNs"])
PY
rm -rf gpurun_out/pn
