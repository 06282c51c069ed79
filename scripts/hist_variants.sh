#!/bin/bash
# development aid: the histogram pass's time on the flat stream for every variant in librir_amd/libs/variants (scripts/variants.py build ...)
set -u
R=$GRAFT_REPO_ROOT
cp $R/librir_amd/libs/librir_amd.so /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
for v in $R/librir_amd/libs/variants/*.so; do
  cp $v $R/librir_amd/libs/librir_amd.so
  n=$(basename $v .so)
  rm -rf /tmp/hv_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/hv_$n -- python3 $R/tests/perf/lossy_flat_time.py 200 7 > /tmp/hv_$n.log 2>&1
  python3 - $n <<PY
import csv,glob,sys
f=glob.glob("/tmp/hv_%s/**/*kernel_trace.csv" % sys.argv[1],recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "lossy_hist_mode_runs_kernel" in r["Kernel_Name"]]
print("%-14s hist us: spread %s | flat %s" % (sys.argv[1], [round(x) for x in d[:6]][-3:], [round(x) for x in d[6:]][-3:]))
PY
done
cp /tmp/keep.so $R/librir_amd/libs/librir_amd.so
