import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.cli_end_to_end as t
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = tempfile.mkdtemp(prefix="bronko_e2e_")
refs, kk, paths = t.prepare(tmp, S, 1000000, NS)
for env_add in ({"BRONKO_INFLATE_THREADS": "1"}, {"BRONKO_INFLATE_THREADS": "2"}, {"BRONKO_INFLATE_THREADS": "4"}, {"BRONKO_INFLATE_THREADS": "8"},
                {"BRONKO_INFLATE_THREADS": "1"}, {"BRONKO_INFLATE_THREADS": "4"}, {"BRONKO_LANES": "8", "BRONKO_INFLATE_THREADS": "8"}, {"BRONKO_LANES": "4", "BRONKO_INFLATE_THREADS": "16"}):
    env = dict(os.environ); env.update(env_add)
    out = os.path.join(tmp, "o")
    t0 = time.time()
    r = subprocess.run([t.BIN, "call", "-g"] + refs + ["-r"] + paths + kk + ["-t", "64", "-o", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(env_add, "%.2f s" % (time.time() - t0), r.returncode, flush=True)
    subprocess.run(["rm", "-rf", out])
subprocess.run(["rm", "-rf", tmp])
