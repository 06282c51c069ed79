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
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/tests/perf/config4_pmc.py > $OUT/$name.log 2> $OUT/$name.err || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/tests/perf/config4_pmc.py > $OUT/stats.log 2> $OUT/stats.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT "lossy_run_kernel|lossy_frame_kernel|lossy_backgrounds" > $OUT/lossy.json
python scripts/pmc_summary.py $OUT "ecc_run_kernel|ecc_run_multi_kernel" > $OUT/ecc.json
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); grep "lossy_run\|ecc_run\|lossy_hist\|lossy_back" $f | cut -d, -f1-4 | cut -c1-200 > $OUT/kernel_avg.csv
cat $OUT/lossy.json $OUT/ecc.json $OUT/kernel_avg.csv
