#!/bin/bash
# PMC passes for the filter kernels of one timing script (default scripts/chain_check.py, timing part only; PMC_SCRIPT=tr_time.py ...).
#   gpurun -- bash scripts/pmc_chain.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_chain
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RIR_CHAIN_TIME_ONLY=1
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/scripts/${PMC_SCRIPT:-chain_check.py} > $OUT/$name.log 2> $OUT/$name.err || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT && python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_chain/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "filter_chain" in k or "gaussian_sep" in k or "translate_" in k or "fix_kernel" in k or "remove_motion" in k:
            acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        print(k, {n: round(sum(v)/len(v)) for n,v in c.items()})
PY
