import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
n, h, w = 4000, 768, 1024   # 6.3 GB of raw frames: offsets beyond 4 GB everywhere
base = torch.from_numpy(s1_noisy_background(250, h, w, seed=3)).cuda()
t = base.repeat(n // 250, 1, 1).contiguous()
t.view(torch.int16)[1::250] += 3  # break exact periodicity a little (uint16 arithmetic through the int16 view)
ctx = D.CodecContext(w, h, n, 50)
enc = ctx.encode(t)
out = ctx.decode(enc)
torch.cuda.synchronize()
ok = torch.equal(out.view(torch.int16), t.view(torch.int16))
t0 = time.perf_counter(); enc = ctx.encode(t); ctx.decode(enc, out=out, check=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("4000 x 1024x768: roundtrip", ok, "ratio %.2f" % (t.numel() * 2 / enc.compressed_bytes()), "%.0f fps" % (n / dt), "raw %.2f TB/s" % (n * 4.0 * h * w / dt / 1e12))
