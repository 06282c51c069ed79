#!/bin/bash
# rocprofv3 PMC passes (caches, memory pipeline, texture addresser) of ecc_run_multi_kernel (tests/perf/ecc_pmc.py: 8 sequences x 64 images in one launch).
# GPU box:  bash scripts/ecc_pmc.sh   -> gpurun_out/ecc_pmc/summary.json
# Every pass runs ONCE, under its own timeout, with the program itself after `--`; its log is kept beside its counters.
# Round 3's note said a TA_* pass "never completed its dispatch".  The log that was kept says otherwise: rocprofiler aborted at the FIRST dispatch of the
# process (gaussian_table_device) with "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect" -
# five TA counters were asked for in one pass and the block does not have that many slots.  The TA counters therefore go one or two to a pass here.
set -u
OUT=${ECC_PMC_OUT:-$GRAFT_REPO_ROOT/gpurun_out/ecc_pmc}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TA_[A-Z0-9_]*" | sort -u > $OUT/ta_counters_available.txt
run() {
  local name=$1; shift
  [ -n "${PASSES:-}" ] && [[ " $PASSES " != *" $name "* ]] && return
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tests/perf/ecc_pmc.py > $OUT/$name.log 2> $OUT/$name.err
  echo "pass $name: exit $?" | tee -a $OUT/passes.txt
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run ta1 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run ta3 TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum
run sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT && python3 - $OUT <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        disp = {}
        for r in csv.DictReader(open(f)):
            if "ecc_run_multi_kernel" not in r["Kernel_Name"]:
                continue
            disp.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
            disp[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for k in disp:
            for c, v in disp[k].items():
                res.setdefault(c, []).append(v)
out = {c: sum(v) / len(v) for c, v in res.items()}
out["_dispatches"] = max((len(v) for v in res.values()), default=0)
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
for n in tcc tcp1 tcp2 ta1 ta2 ta3 sq; do echo "== $n"; tail -2 $OUT/$n.err | cut -c1-200; done
