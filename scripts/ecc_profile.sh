#!/bin/bash
# kernel-trace of the ECC registrator rate script (GPU box): per-kernel average durations and the gaps between kernels
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/ecc_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tests/perf/ecc_time.py 100 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -12
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]  # second repetition
prev_end = None
busy = 0
for r in rows:
    busy += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("second half: %d kernels, span %.1f ms, busy %.1f ms (%.0f %%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
PY
tail -2 $OUT/run.log
