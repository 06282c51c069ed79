#!/bin/bash
# development aid (GPU box): the two forms of the resident loss kernel (RIR_LOSSY_RUN_FORM=5 / 6) on 7, 9, 18 streams, for build variants.  VARIANTS="f1|f2" bash scripts/lossy_forms.sh
IFS="|" read -ra VS <<< "${VARIANTS:-}"
[ ${#VS[@]} -eq 0 ] && VS=("")  # (no variants given: the build as it is)
for v in "${VS[@]}"; do
  touch librir_amd/csrc/lossy_kernels.hip
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for form in 5 6; do
    echo "== variant: [$v] form $form"
    for S in ${STREAMS:-7 9 18}; do RIR_LOSSY_RUN_FORM=$form timeout -k 10 120 python tests/perf/lossy_7_streams.py $S 2>/dev/null | tail -1; done
  done
done
