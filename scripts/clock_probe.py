"""Development aid: encoder time vs the GPU clocks / power the SMU reports while it runs."""
import os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
n, h, w = 1000, 512, 640
t = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
ctx = D.CodecContext(w, h, n, 50)
for _ in range(5): ctx.encode_tiles(t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
p = subprocess.Popen("sleep 0.15; rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | head -8", shell=True, stdout=subprocess.PIPE)
e0.record()
K = 4000
for _ in range(K): ctx.encode_tiles(t)
e1.record(); torch.cuda.synchronize()
print("encode_tiles %.1f us average over %d launches" % (e0.elapsed_time(e1) / K * 1e3, K))
print(p.communicate()[0].decode())
