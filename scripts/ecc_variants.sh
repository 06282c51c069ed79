#!/bin/bash
# development aid (GPU box): rebuild ecc_kernels.hip with other workgroup sizes / counts and time the registrator
#   VARIANTS="flags1|flags2|..." bash scripts/ecc_variants.sh
IFS='|' read -ra VS <<< "${VARIANTS:-|-DRIR_ECC_BLOCK=1024}"
for v in "${VS[@]}"; do
  touch librir_amd/csrc/ecc_kernels.hip librir_amd/csrc/registration_abi.cpp
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== variant: [$v]"
  timeout -k 10 120 python tests/perf/ecc_time.py 2>&1 | grep -v amdgpu.ids
  RIR_ECC_LAUNCH_PER_ITERATION=1 timeout -k 10 120 python tests/perf/ecc_time.py 2>&1 | grep -v amdgpu.ids
done
