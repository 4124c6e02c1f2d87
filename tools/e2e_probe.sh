# Where `bronko call`'s wall time goes with many samples: its log lines stamped as they arrive.
# SAMPLES (32) x 1 M reads, LANES per device (default: the binary's own choice), STRAINS (1; 100 = BASELINE config 5's references)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, re, time, subprocess
sys.path.insert(0, ".")
import tools.cli_end_to_end as t
os.makedirs("/tmp/e2e", exist_ok=True)
S, NS = int(os.environ.get("SAMPLES", "32")), int(os.environ.get("STRAINS", "1"))
refs, kk, paths = t.prepare("/tmp/e2e", S, 1000000, NS)
env = dict(os.environ)
if "LANES" in os.environ: env["BRONKO_LANES"] = os.environ["LANES"]
t0 = time.time()
pr = subprocess.Popen([t.BIN, "call", "-g"] + refs + ["-r"] + paths + kk + ["-t", "64", "-o", "/tmp/e2e/out"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
seen = {}
for line in pr.stdout:
    line = line.decode().rstrip()
    key = re.sub(r"[0-9]+", "#", line.split("/tmp")[0])[:70]
    seen.setdefault(key, []).append(time.time() - t0)
pr.wait()
for k, v in seen.items():
    print("%6.2f .. %6.2f s  x%-3d %s" % (v[0], v[-1], len(v), k))
PY
