#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections, json, os, re
root = sys.argv[1]
want = re.compile(sys.argv[2]) if len(sys.argv) > 2 else re.compile(r"^rirb1")  # kernels to keep (regular expression on the bare name)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"<.*", "", row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rir::", ""))  # (template arguments dropped)
        if not want.search(k):
            continue
        if len(sys.argv) > 3 and sys.argv[3] == "by-grid":  # (launches of one kernel with different grids are different workloads)
            k = "%s@grid%s" % (k, row.get("Grid_Size", "?"))
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in sorted(agg.items()):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    out[k]["_dispatches"] = len(next(iter(cs.values())))
print(json.dumps(out, indent=1))
