# rocprofv3 kernel statistics of one `bronko call` (4 samples x 1 M reads, one lane)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, ".")
import tools.cli_end_to_end as t
os.makedirs("/tmp/e2e", exist_ok=True)
refs, kk, paths = t.prepare("/tmp/e2e", 4, 1000000, 1)
PY
export BRONKO_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2e_prof -- bronko_amd/bin/bronko call -g tests/golden/4_sarscov2/wuhan_ref.fasta -r /tmp/e2e/sample00.fastq.gz /tmp/e2e/sample01.fastq.gz /tmp/e2e/sample02.fastq.gz /tmp/e2e/sample03.fastq.gz -t 8 -o /tmp/e2e/out > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/e2e_prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-60s calls %6s  total %10.1f us  avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/e2e_prof
