#!/bin/bash
# rocprofv3 PMC passes for the resident kernels of configs[4] (tests/perf/config4_pmc.py): lossy_run_kernel, ecc_run_kernel,
# ecc_run_multi_kernel.  GPU box:  bash scripts/pmc_config4.sh <tag>  -> gpurun_out/pmc_config4_<tag>/{lossy,ecc}.json
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_config4_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/tests/perf/config4_pmc.py > $OUT/$name.log 2> $OUT/$name.err || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/tests/perf/config4_pmc.py > $OUT/stats.log 2> $OUT/stats.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_config4_summary.py $OUT $OUT/summary   # -> summary_pmc_lossy.json, summary_pmc_ecc.json (copy to profiles/rNN_pmc_*.json)
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" > $OUT/kernel_avg.csv <<'PY'
import csv, sys
print("kernel,calls,total_ns,average_ns")
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("lossy_run", "ecc_run", "lossy_hist", "lossy_back", "minmax", "gaussian")):
        print('"%s",%s,%s,%.0f' % (n.split("(")[0], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"])))
PY
cat $OUT/kernel_avg.csv
