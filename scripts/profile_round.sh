#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench command + PMC passes.
#   gpurun -- bash scripts/profile_round.sh r01
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
# one stats file per process: bench.py measures its per-frame ABI numbers in a child process (50-frame launches of the same kernels);
# the file of the bench process itself is the one with the most rirb1_decode_tiles time
python - <<PY
import csv, glob, shutil
best, bestt = None, -1
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    t = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(f)) if "rirb1_decode_tiles" in r["Name"])
    if t > bestt:
        best, bestt = f, t
shutil.copy(best, "$OUT/kernel_stats.csv")
print("kernel stats of the bench process:", best)
PY
cd $GRAFT_REPO_ROOT && bash scripts/pmc.sh $TAG > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_$TAG > $OUT/pmc_summary.json
python scripts/pmc_traffic.py $OUT/pmc_summary.json > $OUT/pmc_traffic.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# the constant-budget form of the bounded-loss step (one 640x512 stream, 1 000 frames per call): kernel times and HBM bytes
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lossy_const_stats -- python3 $GRAFT_REPO_ROOT/tests/perf/lossy_const_time.py 1000 1 > $OUT/lossy_const.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/lossy_const_pmc/$c -- python3 $GRAFT_REPO_ROOT/tests/perf/lossy_const_time.py 1000 1 > $OUT/lossy_const_$c.log 2>&1
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/lossy_const_pmc/sq -- python3 $GRAFT_REPO_ROOT/tests/perf/lossy_const_time.py 1000 1 > $OUT/lossy_const_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT/lossy_const_pmc "^lossy_(const|hist|run)" > $OUT/lossy_const_pmc_summary.json
find $OUT/lossy_const_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/lossy_const_kernel_stats.csv
tail -c 600 $OUT/bench.json
