"""Development aid (GPU box): read rate of IRMovie[i] over a 1000-frame recording, per block of 100 frames."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd.synthetic import s1_noisy_background
from librir_amd.video_io import IRMovie, IRSaver
n, h, w = 1000, 512, 640
arr = s1_noisy_background(n, h, w)
d = tempfile.mkdtemp()
for rep in range(3):
    p = os.path.join(d, "a%d.h264" % rep)
    t0 = time.perf_counter()
    with IRSaver(p, w, h, h) as s:
        for i in range(n):
            s.add_image(arr[i], i)
    te = time.perf_counter() - t0
    t0 = time.perf_counter()
    mov = IRMovie.from_filename(p)
    topen = time.perf_counter() - t0
    marks = []
    t0 = time.perf_counter()
    for i in range(n):
        img = mov[i]
        if i % 100 == 99:
            marks.append(time.perf_counter() - t0)
    tr = time.perf_counter() - t0
    t0 = time.perf_counter()
    mov.close()
    tclose = time.perf_counter() - t0
    blocks = np.diff([0] + marks) / 100 * 1e6
    print("rep %d: record %.0f fps; open %.1f ms, read %.0f fps, close %.1f ms; us per frame by block of 100: %s" %
          (rep, n / te, topen * 1e3, n / tr, tclose * 1e3, " ".join("%.0f" % b for b in blocks)))
