#!/usr/bin/env python3
"""Kernel timeline of the last sample in a rocprofv3 kernel trace (run with --in-flight 1: one sample at a time).
usage: sample_timeline.py <kernel_trace.csv> [first kernel of a sample, default pick_window_kernel|scan_count_kernel] [which sample: -1 = the last (default), n = the n-th]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"^bk::", "", n); n = re.sub(r"\(.*$", "", n)
    return n
names = [short(r["Kernel_Name"]) for r in rows]
# a sample starts at the first kernel after a finalize_* / clear_touched / ktab_stats kernel that is none of those
ends = ("finalize_", "clear_touched", "ktab_stats", "select_genome", "noise_", "call_", "copy_int", "prefix_rows", "gather_votes", "gather_table", "voter_table", "merged_votes", "alias_votes", "touched_prows", "counted_ids")
def fill_inside(i):   # a memset between two kernels of a sample's end (the touched-row bits before prefix_rows_kernel)
    j = i
    while j < len(rows) and names[j].startswith("__amd_rocclr_fillBuffer"): j += 1
    return j > i and j < len(rows) and names[j].startswith(("prefix_rows", "gather_", "voter_table"))
starts = [i for i in range(1, len(rows)) if names[i - 1].startswith(ends) and not names[i].startswith(ends) and not fill_inside(i)]
if not starts: sys.exit("no sample boundary found")
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
if which < 0:
    lo = starts[-2] if len(starts) > 1 else 0
    hi = starts[-1]
else:
    lo, hi = starts[which], starts[which + 1]
t0 = int(rows[lo]["Start_Timestamp"])
for i in range(lo, hi):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print("%9.1f us  + %9.1f us  %s" % ((e - s) / 1e3, (s - t0) / 1e3, names[i]))
print("%9.1f us  total" % ((int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))
