#!/bin/bash
# development aid (GPU box): rebuild codec_kernels.hip with diagnostic switches and time the encoder
#   VARIANTS="flags1|flags2|..." CAPS="448 358" bash scripts/enc_variants.sh
IFS='|' read -ra VS <<< "${VARIANTS:-|-DRIR_DIAG_NO_LOOKBACK}"
for v in "${VS[@]}"; do
  touch librir_amd/csrc/codec_kernels.hip
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for c in ${CAPS:-0}; do
    echo "== variant: [$v] cap=$c"
    RIR_SINGLE_PASS=1 RIR_ENC_LDS_WORDS=$c timeout -k 10 120 python tests/perf/enc_ab.py 2>/dev/null | grep -v "slow:\|alone\|segs " | head -${LINES_MAX:-16}
  done
done
