"""Target of the kernel trace of the labelling kernels (GPU box):
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tests/perf/label_prof.py
20 label_image calls each on four 640x512 uint16 images in device memory: empty, regions, vertical stripes of distinct values, one flat
component (profiles/r05_label_kernel_trace.csv)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from librir_amd import device as D
h, w = 512, 640
yy, xx = np.mgrid[0:h, 0:w]
rng = np.random.default_rng(5)
a = np.kron(rng.normal(size=(h // 16 + 2, w // 16 + 2)), np.ones((16, 16)))[:h, :w]
for _ in range(3):
    a = (a + np.roll(a, 5, 0) + np.roll(a, 5, 1) + np.roll(a, -5, 0) + np.roll(a, -5, 1)) / 5
cases = [np.zeros((h, w), np.uint16), (a > 0.4).astype(np.uint16), ((xx % 7) + 1).astype(np.uint16), np.ones((h, w), np.uint16)]
lib = D._lib
for img in cases:
    t = torch.from_numpy(img).cuda()
    need = lib.rir_label_workspace_bytes(w, h)
    work = torch.empty(need // 8 + 1, dtype=torch.int64, device="cuda")
    dst = torch.empty((h, w), dtype=torch.int32, device="cuda")
    xyb = torch.empty((h * w + 1, 2), dtype=torch.float64, device="cuda")
    ab = torch.empty(h * w + 1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    back = np.zeros(1, dtype=img.dtype)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        lib.rir_label_image_device(ord("H"), t.data_ptr(), dst.data_ptr(), w, h, back.ctypes.data, xyb.data_ptr(), ab.data_ptr(), cnt.data_ptr(), work.data_ptr(), work.numel() * 8, st)
    torch.cuda.synchronize()
    torch.zeros(8, device="cuda").sum().item()
