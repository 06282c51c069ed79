#!/bin/bash
# The speculative form of the bounded-loss step on a static 640x512 scene with the reference's default parameters (1 000 frames per call):
# kernel times (rocprofv3 --kernel-trace --stats) and HBM bytes / instruction counters (--pmc, passes of their own).
#   gpurun -- bash scripts/lossy_spec_profile.sh r06      ->  gpurun_out/lossy_spec_<tag>/, summary: scripts/lossy_spec_summary.py
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/lossy_spec_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tests/perf/lossy_spec_time.py
python3 $P 1000 1 > $OUT/rates.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $P 1000 1 speculative,static > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/$c -- python3 $P 1000 1 speculative,static > $OUT/pmc_$c.log 2>&1
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/sq -- python3 $P 1000 1 speculative,static > $OUT/pmc_sq.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cd $GRAFT_REPO_ROOT
python3 scripts/lossy_spec_summary.py $OUT > $OUT/summary.json
cat $OUT/rates.txt
