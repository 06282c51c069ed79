#!/usr/bin/env python3
"""Per-class means of the counters collected by scripts/class_pmc.sh: the dispatches of rirb1_encode_tiles between the last
three rirb1_decode_tiles markers are the 6 launches with the workspace in another class, then the 6 in the same class."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    # one row per (dispatch, counter)
    disp = {}
    for r in rows:
        k = int(r["Dispatch_Id"])
        disp.setdefault(k, {"name": r["Kernel_Name"], "c": {}})
        disp[k]["c"][r["Counter_Name"]] = disp[k]["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = [disp[k] for k in sorted(disp)]
    marks = [i for i, x in enumerate(order) if "rirb1_decode_tiles" in x["name"]]
    if len(marks) < 3:
        continue
    groups = {"other_class": order[marks[-3] + 1:marks[-2]], "same_class": order[marks[-2] + 1:marks[-1]]}
    for g, xs in groups.items():
        xs = [x for x in xs if "rirb1_encode_tiles" in x["name"]][-6:]
        for x in xs:
            for c, v in x["c"].items():
                out.setdefault(c, {}).setdefault(g, []).append(v)
res = {}
for c, gs in out.items():
    res[c] = {g: sum(v) / len(v) for g, v in gs.items()}
    if "other_class" in res[c] and res[c]["other_class"]:
        res[c]["same_over_other"] = res[c]["same_class"] / res[c]["other_class"]
print(json.dumps(res, indent=1))
