cd $GRAFT_REPO_ROOT
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r05_b_bench_default.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r05_b_bench_default.json"))
print("value %.4g  ms/sample %.4f serial %.4f  k0 %.4g" % (d["value"], d["ms_per_sample"], d["serial_ms_per_sample"], d.get("value_with_k0", 0)))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_kernel_ms", "avg_ms_in_flight_incl_queueing")})
print("solo", d["kernels_ms_per_sample_solo"])
for k, o in d.get("other_configs", {}).items():
    print(k, "%.4g" % o["value"], "ms/sample %.3f" % o["ms_per_sample"], "serial %.3f" % o.get("serial_ms_per_sample", 0), o.get("kernels_ms_per_sample_solo"), o.get("setup_s"))
print("cpu", d.get("cpu_baseline"))
print("setup", d["setup_s"])
PY
