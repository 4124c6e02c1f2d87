#!/usr/bin/env python3
"""A window of a rocprofv3 kernel trace as text: every kernel that starts in [mid, mid + span) us of the trace, by start time, with
its queue -- what runs next to what with several samples in flight.   tools/trace_window.py kernel_trace.csv [span us] [frac of the trace]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
scans = [r for r in rows if "scan_items" in r["Kernel_Name"]]   # the window starts at a scan launch in the middle of the run
t0 = scans[int(len(scans) * frac)]["s"] if scans else rows[0]["s"] + int((rows[-1]["e"] - rows[0]["s"]) * frac)
qs = {}
for r in rows:
    if t0 <= r["s"] < t0 + span * 1000:
        q = qs.setdefault(r["Queue_Id"], len(qs))
        name = re.sub(r"^void |bk::|\(.*$|<.*$", "", r["Kernel_Name"])[:28]
        print("%8.1f  %6.1f us  q%d  %s%s  grid %s" % ((r["s"] - t0) / 1000, (r["e"] - r["s"]) / 1000, q, "    " * q, name, r.get("Grid_Size", r.get("Grid_Size_X", ""))))
