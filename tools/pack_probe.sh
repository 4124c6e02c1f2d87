# Host side of `bronko call`'s ingest alone, on the GPU box's cores: one synthetic 4 M-read .fastq.gz inflated (pargz_cat), and
# inflated + parsed + 2-bit packed (pack_cat ... quiet) at several thread counts.   gpurun -- bash tools/pack_probe.sh
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("e2e", "tools/cli_end_to_end.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from bronko_amd import synth
m.write_sample(("/tmp/pp.fastq.gz", 4000000, 101, synth.read_fasta_bytes(m.REF)))
PY
ls -la /tmp/pp.fastq.gz
TIMEFORMAT="%R s wall, %U s user, %S s sys"
for t in 1 16 32 64; do echo -n "pargz_cat $t threads: "; { time bronko_amd/bin/pargz_cat /tmp/pp.fastq.gz $t > /dev/null; } 2>&1; done
for t in 1 8 16 32 64; do echo -n "pack_cat $t threads: "; { time bronko_amd/bin/pack_cat /tmp/pp.fastq.gz 21 $t quiet > /dev/null; } 2>&1; done
