# Is the box able to inflate 32 gzip files side by side?  (zcat of the probe's files, 1 and 32 at a time)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time, subprocess
sys.path.insert(0, ".")
import tools.cli_end_to_end as t
os.makedirs("/tmp/e2e", exist_ok=True)
from multiprocessing import Pool
paths = ["/tmp/e2e/s%02d.fastq.gz" % i for i in range(32)]
with Pool(16) as p: p.map(t.write_sample, [(q, 1000000, 300 + i) for i, q in enumerate(paths)])
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for n in (1, 8, 32):
    t0 = time.time()
    ps = [subprocess.Popen("zcat %s | wc -c > /dev/null" % q, shell=True) for q in paths[:n]]
    for p in ps: p.wait()
    print("%d zcat side by side: %.2f s" % (n, time.time() - t0), flush=True)
PY
