#!/bin/bash
# development aid (GPU box): where the time of a frame goes in the bounded-loss run kernel (build with -DRIR_LOSSY_DIAG)
touch librir_amd/csrc/lossy_kernels.hip
RIR_EXTRA_CFLAGS="-DRIR_LOSSY_DIAG ${EXTRA:-}" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for S in ${STREAMS:-1 6}; do
  RIR_LOSSY_DIAG=1 timeout -k 10 120 python tests/perf/lossy_soak.py $S 60 20 2>&1 | grep -v amdgpu.ids | tail -2
done
