#!/bin/bash
# rocprofv3 PMC passes of the packing kernel with its workspace in another / in the same placement class as the frames
# (tests/perf/class_pmc.py).  GPU box:  bash scripts/class_pmc.sh   -> gpurun_out/class_pmc/<pass>/...counter_collection.csv
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/class_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/tests/perf/class_pmc.py > $OUT/$name.log 2> $OUT/$name.err || echo "pass $name failed"
  tail -2 $OUT/$name.log
}
run ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE
run ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum
run tcc TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
cd $GRAFT_REPO_ROOT && python scripts/class_pmc_summary.py $OUT > $OUT/summary.json && cat $OUT/summary.json
