cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BK_L2_STATS=1 python tools/many_strains_check.py 30 200000 2>&1 | grep -v "^\[bk\] scan" | tail -14
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ms -- python3 tools/many_strains_check.py 30 200000 --no-oracle > gpurun_out/prof_ms.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_ms/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bk::" in r["Name"]: print(r["Name"][:60].ljust(60), r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/prof_ms
