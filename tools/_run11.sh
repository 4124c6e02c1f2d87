cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/nz.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from bronko_amd import _ffi
_ffi.use_testing_library(True)
from tests import helpers
from oracle import oracle as orc
from bronko_amd import synth
orc.build()
p = os.path.join("tests", "golden", "4_sarscov2", "wuhan_ref.fasta")
ix = orc.Index.build(21, [p])
eng = helpers.engine_from_oracle_index(ix)
gm, isnv = synth.sample_genome(synth.read_fasta_bytes(p), 41)
reads = synth.codes_to_ascii(synth.single_end_codes(gm, 400000, 150, 41, isnv=isnv))
for rep in range(3):
    helpers.hip_sample(eng, [reads], 21)
    eng.sample_call(1, eng.call_params())
    eng.download_calls()
PY
for m in 0 3; do
BK_NOISE_SERIAL=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pn -- python3 /tmp/nz.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/pn/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "noise_walk" in r["Name"]: print("mode $m", r["Name"][:40], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/pn
done
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_calls.py -x -q -m gpu 2>&1 | tail -3
