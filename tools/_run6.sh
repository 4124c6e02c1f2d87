cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in rel test; do
if [ $ab = test ]; then export BK_L2_STATS=1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ms -- python3 tools/many_strains_check.py 30 200000 --no-oracle > gpurun_out/prof_ms.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof_ms/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gather_votes" in r["Name"] or "prefix" in r["Name"]: print("$ab", r["Name"][:50].ljust(50), r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3), r["MinNs"], r["MaxNs"])
PY
rm -rf gpurun_out/prof_ms
done
