#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench command + PMC passes.
#   gpurun -- bash scripts/profile_round.sh r01
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cd $GRAFT_REPO_ROOT && bash scripts/pmc.sh $TAG > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_$TAG > $OUT/pmc_summary.json
python scripts/pmc_traffic.py $OUT/pmc_summary.json > $OUT/pmc_traffic.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
