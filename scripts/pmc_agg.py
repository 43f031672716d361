#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from the counter_collection CSVs under a directory.
A kernel that is launched over very different amounts of work (the candidate filter: once over the bound's sample, once
over the whole index) is reported per class: dispatches whose first counter is within a factor of four of the kernel's
largest form the class `name`, the others `name [small]`."""
import csv, glob, json, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(dict))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        d = (f, r["Dispatch_Id"])
        acc[name][d][r["Counter_Name"]] = acc[name][d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
out = {}
for name, disp in acc.items():
    first = sorted(next(iter(disp.values())).keys())[0]
    top = max(v.get(first, 0.0) for v in disp.values())
    groups = {name: [], name + " [small]": []}
    for v in disp.values():
        groups[name if top == 0 or v.get(first, 0.0) * 4 >= top else name + " [small]"].append(v)
    for gname, vs in groups.items():
        if not vs:
            continue
        cs = sorted({c for v in vs for c in v})
        out[gname] = {c: sum(v.get(c, 0.0) for v in vs) / len(vs) for c in cs} | {"dispatches": len(vs)}
print(json.dumps(out, indent=1))
