#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: per kernel, average duration and how much of it ran while a kernel of another
stream (queue) was running too; wall time per bench step.  Usage: trace_overlap.py kernel_trace.csv [steps [warmup]]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bk::", "").replace("bk::", ""), r.get("Queue_Id", "0")) for r in rows]
ev.sort()
# bench.py's timed region: the scan launches after the warm-up ones, `steps` of them (default: 3 warm-up, 20 timed)
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scans = [e for e in ev if e[2].startswith(("scan_count", "scan_items"))]
t_lo = scans[warm][0]
t_hi = scans[warm + steps][0] if len(scans) > warm + steps else ev[-1][1]
ev = [e for e in ev if t_lo <= e[0] < t_hi]
tot = defaultdict(int); cnt = defaultdict(int); ovl = defaultdict(int)
for i, (s, e, n, q) in enumerate(ev):
    tot[n] += e - s; cnt[n] += 1
    o = 0
    for (s2, e2, n2, q2) in ev[max(0, i - 40):i + 40]:
        if q2 != q and s2 < e and e2 > s:
            o += min(e, e2) - max(s, s2)
    ovl[n] += min(o, e - s)
wall = ev[-1][1] - ev[0][0]
nscan = sum(1 for e in ev if e[2].startswith(("scan_count", "scan_items")))
print("steady-state window: %.3f ms, %d scan launches -> %.4f ms per step" % (wall / 1e6, nscan, wall / 1e6 / max(nscan, 1)))
for n in sorted(tot, key=lambda x: -tot[x]):
    print("%-40s n=%4d avg %8.1f us   overlapped with another queue's kernel: %4.0f %%" % (n[:40], cnt[n], tot[n] / cnt[n] / 1e3, 100.0 * ovl[n] / tot[n]))
print("sum of kernel time per step: %.4f ms" % (sum(tot.values()) / 1e6 / max(nscan, 1)))
