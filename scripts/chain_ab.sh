#!/bin/bash
# development aid (GPU box): the fused filter chain as one kernel (round 2) against the regular / listed split, same box
for v in "-DRIR_CHAIN_ONE_KERNEL" "" "-DRIR_CHAIN_ONE_KERNEL" ""; do
  touch librir_amd/csrc/filter_kernels.hip
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== variant: [$v]"
  python scripts/bench_filters.py 2>/dev/null | grep "filter_chain\|translate u16 nearest"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/chain_split_prof -- python $GRAFT_REPO_ROOT/scripts/bench_filters.py > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/chain_split_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_chain" in r["Name"] or "bad_pixels_fix" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
