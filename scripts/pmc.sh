#!/bin/bash
# rocprofv3 PMC passes for the codec kernels of the bench step (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2); `--profile`: the step
# alone - no ramp, no repeats, none of the other forms (under --pmc every dispatch is serialised and costs milliseconds).
# Run on the GPU box through gpurun:  bash scripts/pmc.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-abi --profile > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
find $OUT -name "*counter_collection.csv" | head
