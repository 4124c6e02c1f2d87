# which kernels a sample's time goes to at N strains (default 250): rocprofv3 kernel statistics of tools/scale_probe.py N 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-250}
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sk_$N -- python3 tools/scale_probe.py $N 4 > gpurun_out/sk_$N.log 2>&1
grep -v amdgpu.ids gpurun_out/sk_$N.log | tail -6 | cut -c1-200
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/sk_$N/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg %9.1f us" % (float(r["AverageNs"]) / 1e3), "total %8.1f ms" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
PY
rm -rf gpurun_out/sk_$N
