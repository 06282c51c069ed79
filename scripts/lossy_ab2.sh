#!/bin/bash
# development aid (GPU box): build variants of the resident bounded-loss kernel and time 7 / 8 / 14 / 16 streams.   VARIANTS="f1|f2" bash scripts/lossy_ab2.sh
IFS='|' read -ra VS <<< "${VARIANTS:-|-DRIR_LOSSY_LEADER_PRIO}"
for v in "${VS[@]}"; do
  touch librir_amd/csrc/lossy_kernels.hip
  RIR_EXTRA_CFLAGS="$v" python -c "from librir_amd import build; build.build(verbose=False)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== variant: [$v]"
  timeout -k 10 300 python - <<'PY' 2>/dev/null
import time, torch, sys
sys.path.insert(0, ".")
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
h, w, m = 512, 640, 200
fr = torch.from_numpy(s1_noisy_background(m, h, w)).cuda()
for S in [int(x) for x in __import__("os").environ.get("STREAMS", "7,8,14,16").split(",")]:
    streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
    ins = [fr.clone() for _ in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)
    best = 0
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        D.LossyStream.step_many(streams, ins, errors=False)
        torch.cuda.synchronize(); best = max(best, m * S / (time.perf_counter() - t0))
    st = streams[0].status()
    print("%2d streams x 200 frames: %.0f k fps aggregate (status %s)" % (S, best / 1e3, st))
    for x in streams: x.close()
PY
done
