#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries of tools/profile_round.sh: the scan kernel's HBM bytes and VALU
wave-instructions per launch, stamped with the id of the sources they were measured on (bench.py reports the figure only for
that build).  usage: make_pmc_traffic.py <pmc_hbm.json> <pmc_sq.json> <capture tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

hbm = json.load(open(sys.argv[1]))
sq = json.load(open(sys.argv[2]))
# The correction factors are MEASURED (tools/micro/fetch_calibrate.hip -> profiles/r04_fetch_calibration.json): FETCH_SIZE tallies
# every 128-byte request as 64 bytes whatever the access pattern -- streaming 16-, 8- and 4-byte-per-lane loads all report exactly
# half of what they read, a scattered 8- or 16-byte load reports 64 bytes for its 128-byte request -- so traffic = 2 x FETCH_SIZE
# for every kernel here; WRITE_SIZE is exact for streaming stores and 32 bytes per scattered 8-byte atomic (the sector written
# back), i.e. it is the traffic as it stands.  pmc_summary.py applies exactly these two factors (x 2, x 1).
cal = os.path.join(ROOT, "profiles", "r04_fetch_calibration.json")
scan_h = next(v for k, v in hbm.items() if "scan_items" in k or "scan_count" in k)
scan_s = next(v for k, v in sq.items() if "scan_items" in k or "scan_count" in k)
out = {"build_id": bench.source_build_id(), "config": 2, "workload_reads": 1000000, "read_len": 150, "kernel": next(k for k in hbm if "scan_items" in k or "scan_count" in k).split("(")[0].split("::")[-1].split("<")[0],
       "fetch_write_calibration": os.path.relpath(cal, ROOT) if os.path.exists(cal) else None,
       "hbm_read_bytes_per_launch": scan_h["hbm_read_bytes"], "hbm_write_bytes_per_launch": scan_h["hbm_write_bytes"],
       "traffic_bytes_per_launch": scan_h["hbm_read_bytes"] + scan_h["hbm_write_bytes"],
       "valu_wave_insts_per_launch": scan_s.get("SQ_INSTS_VALU"), "salu_wave_insts_per_launch": scan_s.get("SQ_INSTS_SALU"),
       "lds_wave_insts_per_launch": scan_s.get("SQ_INSTS_LDS"), "waves_per_launch": scan_s.get("SQ_WAVES"),
       "captured": "gpurun_out/%s_pmc_hbm.json / _pmc_sq.json (tools/profile_round.sh %s, one sample at a time)" % (sys.argv[3], sys.argv[3]),
       "all_kernels_hbm_bytes_per_launch": {k.split("(")[0][-60:]: v.get("hbm_read_bytes", 0) + v.get("hbm_write_bytes", 0) for k, v in hbm.items()}}
json.dump(out, sys.stdout, indent=1)
print()
