import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 14 kernels
last=rows[-16:]
t0=int(last[0]["Start_Timestamp"])
prev_end=None
for r in last:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
    gap = (s-prev_end) if prev_end is not None else 0
    print("%9.1f us  +gap %6.1f  dur %8.1f  %s" % (s/1e3, gap/1e3, (e-s)/1e3, r["Kernel_Name"][:60]))
    prev_end=e
