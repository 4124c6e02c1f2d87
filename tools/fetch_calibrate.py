#!/usr/bin/env python3
"""factor = true bytes / bytes rocprofv3 reports (FETCH_SIZE / WRITE_SIZE, KiB) for each pattern of tools/micro/fetch_calibrate.hip.
usage: fetch_calibrate.py <stdout of the probe> <counter_collection.csv> ..."""
import collections, csv, json, sys
true = {}
for line in open(sys.argv[1]):
    if line.startswith("true bytes:"):
        w = line.split()[2:]
        true = {w[i]: int(w[i + 1]) for i in range(0, len(w), 2)}
rep = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].split()[-1]
        rep[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "second repetition of each kernel; factor = true bytes / (counter * 1024)", "patterns": {}}
for name, t in true.items():
    d = {"true_bytes": t}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = rep.get(name, {}).get(c)
        if v:
            d[c + "_bytes"] = v[-1] * 1024
            d[c.split("_")[0].lower() + "_factor"] = (t / (v[-1] * 1024)) if v[-1] else None
    out["patterns"][name] = d
json.dump(out, sys.stdout, indent=1)
print()
