#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from the counter_collection CSVs under a directory."""
import csv, glob, json, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    for (d, c), v in per_dispatch.items():
        acc[names[d]][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(next(iter(cs.values())))} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
