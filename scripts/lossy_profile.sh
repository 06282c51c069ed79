#!/bin/bash
# kernel-trace of the bounded-loss rates script (GPU box): per-kernel average durations
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/lossy_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tests/perf/lossy_time.py > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-7 $f | head -12
