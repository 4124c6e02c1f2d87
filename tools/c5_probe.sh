# config 5 (100 strains, k = 31) between two builds: the many-genome parity tests, one sample's kernels in order (every genome's
# rows), instruction counts of the larger kernels, the bench line.   gpurun -- bash tools/c5_probe.sh [notest]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ "$1" = notest ] || timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "config5_hundred or every_genomes_rows_by_table or file_bitmaps or many_strains_k31 or multi_sequence" 2>&1 | tail -5
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python3 bench.py --config 5 --in-flight 1 --steps 1 --warmup 1 --samples-per-step 2 --no-cpu-baseline > /dev/null 2>&1
python3 tools/sample_timeline.py $(find gpurun_out/tl5 -name "*kernel_trace.csv") > gpurun_out/c5_timeline.txt
rm -rf gpurun_out/tl5
grep -v "fill\|zero_small\|compact\|expand\|choose\|pick\|alias\|merged\|clear_t\|reduce" gpurun_out/c5_timeline.txt | cut -c1-100
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc5_q -- python3 bench.py --config 5 --in-flight 1 --steps 1 --warmup 1 --samples-per-step 2 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/pmc5_q -name "*counter_collection.csv") > gpurun_out/c5_sq.json
rm -rf gpurun_out/pmc5_q
python3 - <<PY
import json
d = json.load(open("gpurun_out/c5_sq.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:9]:
    if v["launches"] > 1: print(" ", k[:44].ljust(44), {a: round(b / 1e6, 2) for a, b in v.items() if a != "launches"}, v["launches"])
PY
python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
