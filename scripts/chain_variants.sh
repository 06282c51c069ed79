#!/bin/bash
# development aid (GPU box): filter_chain store policy variants - time (bench_filters.py) and HBM write bytes (one PMC pass)
#   VARIANTS="flags1|flags2" bash scripts/chain_variants.sh
IFS="|" read -ra VS <<< "${VARIANTS:-|-DRIR_CHAIN_STORE_AUX=2}"
for v in "${VS[@]}"; do
  touch librir_amd/csrc/filter_kernels.hip
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== variant: [$v]"
  python scripts/bench_filters.py 2>/dev/null | grep -i "chain\|median"
  ( cd /tmp && export TMPDIR=/tmp && RIR_CHAIN_TIME_ONLY=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw_$$ -- python3 $GRAFT_REPO_ROOT/scripts/chain_check.py > /dev/null 2>&1;
    python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/pmcw_$$/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "filter_chain" in k or "median3x3" in k:
            acc[k.split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("  WRITE_SIZE %-40s %.1f MB per dispatch (%d dispatches)"%(k, sum(v)/len(v)*1024/1e6, len(v)))
PY
    rm -rf /tmp/pmcw_$$ )
done
