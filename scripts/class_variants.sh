#!/bin/bash
# development aid (GPU box): rebuild codec_kernels.hip with cache-policy switches and time the packing kernel with its
# workspace in the same / in another placement class as the frames (tests/perf/class_probe.py)
#   VARIANTS="flags1|flags2|..." bash scripts/class_variants.sh
IFS='|' read -ra VS <<< "${VARIANTS:-|-DRIR_SPARSE_STORE_AUX=2}"
for v in "${VS[@]}"; do
  touch librir_amd/csrc/codec_kernels.hip
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== variant: [$v]"
  timeout -k 10 120 python tests/perf/class_probe.py 2>/dev/null | tail -1
done
