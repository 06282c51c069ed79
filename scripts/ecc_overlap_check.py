import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
multi = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "ecc_run_multi" in r["Kernel_Name"]]
prep = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30]) for r in rows if any(k in r["Kernel_Name"] for k in ("minmax_", "gaussian"))]
inside = sum(1 for p in prep if any(m[0] <= p[0] and p[1] <= m[1] for m in multi))
print("multi launches %d, pre-processing kernels %d, of which entirely inside a multi launch: %d" % (len(multi), len(prep), inside))
for m in multi[:6]:
    ins = [p for p in prep if m[0] <= p[0] <= m[1]]
    if ins:
        print(" multi %.0f us: %d kernels started inside, first at +%.0f us, last ended at +%.0f us; their busy time %.0f us" % ((m[1]-m[0])/1e3, len(ins), (ins[0][0]-m[0])/1e3, (max(p[1] for p in ins)-m[0])/1e3, sum(p[1]-p[0] for p in ins)/1e3))
    else:
        print(" multi %.0f us: none inside" % ((m[1]-m[0])/1e3))
