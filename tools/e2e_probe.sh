# Where `bronko call`'s wall time goes with many samples: its log lines stamped as they arrive (32 x 1 M reads, 32 lanes)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time, subprocess
sys.path.insert(0, ".")
import tools.cli_end_to_end as t
os.makedirs("/tmp/e2e", exist_ok=True)
from multiprocessing import Pool
paths = ["/tmp/e2e/s%02d.fastq.gz" % i for i in range(32)]
with Pool(16) as p: p.map(t.write_sample, [(q, 1000000, 300 + i) for i, q in enumerate(paths)])
env = dict(os.environ, BRONKO_LANES=os.environ.get("LANES", "32"))
t0 = time.time()
pr = subprocess.Popen([t.BIN, "call", "-g", t.REF, "-r"] + paths + ["-t", "64", "-o", "/tmp/e2e/out"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
seen = {}
for line in pr.stdout:
    line = line.decode().rstrip()
    import re
    key = re.sub(r"[0-9]+", "#", line.split("/tmp")[0])[:70]
    seen.setdefault(key, []).append(time.time() - t0)
pr.wait()
for k, v in seen.items():
    print("%6.2f .. %6.2f s  x%-3d %s" % (v[0], v[-1], len(v), k))
PY
