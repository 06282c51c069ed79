"""Development aid: one filter in a loop, for rocprofv3 --pmc.   python scripts/filter_loop.py median|translate|gaussian|remove_motion"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
which = sys.argv[1] if len(sys.argv) > 1 else "median"
t = torch.from_numpy(s1_noisy_background(256, 512, 640)).cuda()
f = t.to(torch.float32)
sh = torch.tensor([[1.25, -2.5]] * 256, dtype=torch.float32, device="cuda")
fn = {"median": lambda: D.median_filter(t), "translate": lambda: D.translate(t, (1.25, -2.5), "nearest"),
      "gaussian": lambda: D.gaussian_filter(f, 0.75), "remove_motion": lambda: D.remove_motion(t, sh, rows=509)}[which]
for _ in range(6):
    fn()
torch.cuda.synchronize()
