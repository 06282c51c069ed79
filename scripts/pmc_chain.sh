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
import csv,glob,collections,json
out={}
for f in sorted(glob.glob("gpurun_out/pmc_chain/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "rir::" in k and ("filter_chain" in k or "gaussian_sep" in k or "translate_" in k or "fix_kernel" in k or "median3x3" in k or "clamp_copy" in k):
            acc[k.split("(")[0].replace("void ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        out.setdefault(k,{}).update({n: round(sum(v)/len(v)) for n,v in c.items()})
for k,c in out.items():
    if "FETCH_SIZE" in c: c["hbm_fetch_MB"]=round(c["FETCH_SIZE"]*1024*2/1e6,1)   # gfx950 correction: x2
    if "WRITE_SIZE" in c: c["hbm_write_MB"]=round(c["WRITE_SIZE"]*1024/1e6,1)
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c: c["valu_per_wave"]=round(c["SQ_INSTS_VALU"]/c["SQ_WAVES"],1)
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c: c["valu_busy_frac"]=round(c["SQ_INSTS_VALU"]*4/1024/(c["GRBM_GUI_ACTIVE"]/8),3)
json.dump({"note":"rocprofv3 --pmc passes (SQ / GRBM / FETCH_SIZE / WRITE_SIZE, each in its own run), mean per dispatch, counters summed over the 8 XCDs; valu_busy_frac = VALU instructions x 4 cycles / 1024 SIMDs / kernel cycles","kernels":out}, open("gpurun_out/pmc_chain/summary.json","w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
