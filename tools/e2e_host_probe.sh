# Is the box able to inflate 32 gzip files side by side?  (zcat of the probe's files, 1 and 32 at a time)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time, subprocess
sys.path.insert(0, ".")
import tools.cli_end_to_end as t
os.makedirs("/tmp/e2e", exist_ok=True)
refs, kk, paths = t.prepare("/tmp/e2e", 32, 1000000, 1)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for n in (1, 8, 32):
    t0 = time.time()
    ps = [subprocess.Popen("zcat %s | wc -c > /dev/null" % q, shell=True) for q in paths[:n]]
    for p in ps: p.wait()
    print("%d zcat side by side: %.2f s" % (n, time.time() - t0), flush=True)
PY
