#!/bin/bash
# development aid (GPU box): rebuild ecc_kernels.hip with (pixels per round, waves per SIMD) of the multi-sequence kernel and time it
#   VARIANTS="flags1|flags2" bash scripts/ecc_multi_variants.sh
IFS='|' read -ra VS <<< "${VARIANTS:-|-DRIR_ECC_MULTI_PIXELS_PER_ROUND=2}"
for v in "${VS[@]}"; do
  touch librir_amd/csrc/ecc_kernels.hip
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== variant: [$v]"
  timeout -k 10 200 python tests/perf/ecc_multi_time.py ${S:-8} 100 2>&1 | grep "^S=\|^breakdown\|^ecc multi" | tail -3
done
