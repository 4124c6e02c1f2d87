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
scan_h = next(v for k, v in hbm.items() if "scan_count" in k)
scan_s = next(v for k, v in sq.items() if "scan_count" in k)
out = {"build_id": bench.source_build_id(), "config": 2, "workload_reads": 1000000, "read_len": 150, "kernel": "scan_count_kernel",
       "hbm_read_bytes_per_launch": scan_h["hbm_read_bytes"], "hbm_write_bytes_per_launch": scan_h["hbm_write_bytes"],
       "traffic_bytes_per_launch": scan_h["hbm_read_bytes"] + scan_h["hbm_write_bytes"],
       "valu_wave_insts_per_launch": scan_s.get("SQ_INSTS_VALU"), "salu_wave_insts_per_launch": scan_s.get("SQ_INSTS_SALU"),
       "lds_wave_insts_per_launch": scan_s.get("SQ_INSTS_LDS"), "waves_per_launch": scan_s.get("SQ_WAVES"),
       "captured": "gpurun_out/%s_pmc_hbm.json / _pmc_sq.json (tools/profile_round.sh %s, one sample at a time)" % (sys.argv[3], sys.argv[3]),
       "all_kernels_hbm_bytes_per_launch": {k.split("(")[0][-60:]: v.get("hbm_read_bytes", 0) + v.get("hbm_write_bytes", 0) for k, v in hbm.items()}}
json.dump(out, sys.stdout, indent=1)
print()
