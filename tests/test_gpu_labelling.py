"""label_image / keep_largest_area on the GPU (csrc/label_kernels.hip) through the C ABI: against the golden vectors made from the
compiled reference, against the oracle on random images, and - at the BASELINE geometries - through properties that do not need either.
Reference tests: tests/python/test_rir.py:279-300 (the wrappers' error behaviour)."""
import numpy as np
import pytest

from cases import LABEL_DTYPES, label_cases
from test_labelling_oracle import check_against_golden, label_golden  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


class _Wrapper:
    """librir_amd.signal_processing with the oracle's method names"""

    def __init__(self):
        from librir_amd import signal_processing as sp

        self.sp = sp

    def label_image(self, img, bg=0):
        return self.sp.label_image(img, bg)

    def keep_largest_area(self, img, bg=0, fg=1):
        return self.sp.keep_largest_area(img, bg, fg)


@pytest.fixture(scope="module")
def wrapper(dev):
    return _Wrapper()


def test_every_golden_case_through_the_c_abi(wrapper, label_golden):  # noqa: F811
    assert check_against_golden(wrapper, label_golden) > 600


def test_random_images_against_the_oracle(wrapper, oracle):
    rng = np.random.default_rng(11)
    for it in range(400):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 300))
        dt = LABEL_DTYPES[it % len(LABEL_DTYPES)]
        levels = int(rng.integers(1, 6))
        img = rng.integers(0, levels + 1, (h, w)).astype(dt)
        if it % 3 == 0:  # larger structures
            img = np.kron(img[:(h + 3) // 4, :(w + 3) // 4], np.ones((4, 4), dtype=img.dtype))[:h, :w].astype(dt)
        if np.dtype(dt).kind == "f" and it % 2 == 0:
            img[rng.random((h, w)) < 0.03] = np.nan
        bg = int(rng.integers(0, 2))
        got, exp = wrapper.label_image(img, bg), oracle.label_image(img, bg)
        for a, b in zip(got, exp):
            assert np.array_equal(a, b), (it, h, w, dt)
        fg = int(rng.integers(-5, 6))
        assert np.array_equal(wrapper.keep_largest_area(img, bg, fg), oracle.keep_largest_area(img, bg, fg)), (it, h, w, dt)


def test_device_layer_against_the_oracle(dev, oracle):
    import torch

    for dt in (np.uint8, np.uint16, np.int32, np.int64, np.float32, np.float64):
        for name, img, bg in label_cases(67, 83, dt) + label_cases(23, 64, dt):
            t = torch.from_numpy(img).cuda()
            lab, area, xy = dev.label_image(t, bg)
            exp = oracle.label_image(img, bg)
            assert np.array_equal(lab.cpu().numpy(), exp[0]) and np.array_equal(area.cpu().numpy(), exp[1]), (dt, name)
            assert np.array_equal(xy.cpu().numpy(), exp[2]), (dt, name)
            assert np.array_equal(dev.keep_largest_area(t, bg, -2).cpu().numpy(), oracle.keep_largest_area(img, bg, -2)), (dt, name)


@pytest.mark.parametrize("shape", [(512, 640), (768, 1024), (1537, 2049)])
def test_properties_at_full_size(wrapper, shape):
    """what the labels must satisfy whatever produced them: each table entry counts its label's pixels and holds its first pixel's x,
    labels rise with the first pixels, labelling the label image changes nothing, vertical neighbours of components share a label,
    horizontal neighbours share one exactly when their values are equal"""
    h, w = shape
    rng = np.random.default_rng(h)
    small = rng.integers(0, 4, ((h + 7) // 8, (w + 7) // 8))
    img = np.kron(small, np.ones((8, 8), dtype=np.int64))[:h, :w]
    img[rng.random((h, w)) < 0.02] = 5
    img = img.astype(np.uint16)
    lab, area, xy = wrapper.label_image(img, 0)
    n = area.size
    assert n >= 2 and lab.min() == 0 and lab.max() == n - 1
    assert np.array_equal((lab == 0), (img == 0))
    counts = np.bincount(lab.ravel(), minlength=n)
    assert np.array_equal(counts[1:], area[1:]) and area[0] == 0
    flat = lab.ravel()
    first = np.full(n, flat.size, dtype=np.int64)
    np.minimum.at(first, flat, np.arange(flat.size))
    assert np.all(np.diff(first[1:]) > 0)  # numbered in raster order of the first pixels
    assert np.array_equal(xy[1:, 0], first[1:] % w) and np.array_equal(xy[1:, 1], xy[1:, 0]) and np.array_equal(xy[0], [-1, -1])
    both = (img[1:] != 0) & (img[:-1] != 0)
    assert np.array_equal(lab[1:][both], lab[:-1][both])
    side = (img[:, 1:] != 0) & (img[:, :-1] != 0)
    same = img[:, 1:] == img[:, :-1]
    assert np.all(lab[:, 1:][side & same] == lab[:, :-1][side & same])
    again, area2, xy2 = wrapper.label_image(lab, 0)
    assert np.array_equal(again, lab) and np.array_equal(area2, area) and np.array_equal(xy2, xy)
    keep = wrapper.keep_largest_area(img, 0, 3)
    best = 1 + int(np.argmax(area[1:]))  # argmax: the first among equals
    assert np.array_equal(keep, np.where(lab == best, 3, 0))


def test_degenerate_images_and_errors(wrapper, lib):
    import ctypes as ct

    from librir_amd.signal_processing import keep_largest_area, label_image

    lab, area, xy = label_image(np.zeros((0, 5), np.uint16), 0)
    assert lab.shape == (0, 5) and np.array_equal(area, [0]) and np.array_equal(xy, [[-1, -1]])
    assert keep_largest_area(np.zeros((3, 0), np.uint8), 0).shape == (3, 0)
    # every pixel of a one-row image its own component: w + 1 table entries
    row = (np.arange(70) % 2 + 1).astype(np.uint8).reshape(1, 70)
    lab, area, xy = label_image(row, 0)
    assert np.array_equal(lab, np.arange(1, 71).reshape(1, 70)) and area.size == 71 and np.all(area[1:] == 1)
    # reference tests/python/test_rir.py:279-300
    with pytest.raises(RuntimeError):
        label_image(np.ndarray((10, 10, 10)), 0)
    with pytest.raises(RuntimeError):
        label_image(np.ndarray((10, 10), dtype="object"), 0)
    with pytest.raises(RuntimeError):
        keep_largest_area(np.ndarray((10, 10, 10)), 0)
    with pytest.raises(RuntimeError):
        keep_largest_area(np.ndarray((10, 10), dtype="object"), 0)
    # the C entry points: unknown type character
    img = np.ones((4, 4), np.uint16)
    dst = np.zeros((4, 4), np.int32)
    bg = np.zeros(1, np.uint16)
    xyb, ab = np.zeros(40), np.zeros(20, np.int32)
    lib.label_image.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p]
    lib.keep_largest_area.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int]
    assert lib.label_image(ord("O"), img.ctypes.data, dst.ctypes.data, 4, 4, bg.ctypes.data, xyb.ctypes.data, ab.ctypes.data) == -1
    assert lib.keep_largest_area(ord("x"), img.ctypes.data, dst.ctypes.data, 4, 4, bg.ctypes.data, 1) == -1
    assert lib.label_image(ord("H"), img.ctypes.data, dst.ctypes.data, 4, 4, bg.ctypes.data, xyb.ctypes.data, ab.ctypes.data) == 2


def test_labelling_on_the_copy_path(oracle):
    """RIR_ABI_ZERO_COPY=0: the labels come back through a device buffer and a transfer instead of being written over the link"""
    import os
    import subprocess
    import sys

    code = ("import numpy as np, sys\n"
            "sys.path.insert(0, %r)\n"
            "from librir_amd import signal_processing as sp\n"
            "rng = np.random.default_rng(3)\n"
            "img = np.kron(rng.integers(0, 3, (40, 50)), np.ones((8, 8), dtype=np.int64)).astype(np.uint16)\n"
            "lab, area, xy = sp.label_image(img, 0)\n"
            "np.savez(sys.argv[1], lab=lab, area=area, xy=xy, keep=sp.keep_largest_area(img, 0, 4), img=img)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.npz")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, RIR_ABI_ZERO_COPY="0"))
        z = np.load(out)
        exp = oracle.label_image(z["img"], 0)
        assert np.array_equal(z["lab"], exp[0]) and np.array_equal(z["area"], exp[1]) and np.array_equal(z["xy"], exp[2])
        assert np.array_equal(z["keep"], oracle.keep_largest_area(z["img"], 0, 4))


def test_the_reference_scenario(wrapper):
    """reference tests/python/test_rir.py:306-314: two filled shapes of one value on an int32 image (drawn there with the geometry
    library the drop-in keeps; rasterised by hand here), labelled and reduced to the larger one with the default arguments"""
    from librir_amd import signal_processing as sp

    img = np.zeros((20, 20), np.int32)
    for y in range(10):
        img[y, 0:max(1, 6 - y // 2)] = 5  # a wedge from the corner
    img[15:18, 14:17] = 5  # a small block
    lab, area, xy = sp.label_image(img)
    assert area.size == 3 and lab[0, 0] == 1 and lab[16, 15] == 2 and area[1] == (img[:10] == 5).sum() and area[2] == 9
    assert np.array_equal(xy, [[-1, -1], [0, 0], [14, 14]])
    keep = sp.keep_largest_area(img)
    assert keep.dtype == np.int32 and np.array_equal(keep, (lab == 1).astype(np.int32))


@pytest.mark.perf
def test_rate_floor_of_the_labelling(dev):
    """a 640x512 image of a few regions: the five launches on an image in device memory well under 100 us (measured 24-32), the C entry
    point under 400 us per host image (measured 100-140; the reference's scan takes 0.8-3 ms)"""
    import time

    import torch

    from librir_amd import signal_processing as sp

    rng = np.random.default_rng(5)
    img = np.kron(rng.integers(0, 3, (32, 40)), np.ones((16, 16), dtype=np.int64)).astype(np.uint16)
    t = torch.from_numpy(img).cuda()
    best_dev = best_abi = 1e9
    for rep in range(5):
        dev.keep_largest_area(t, 0, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            out = dev.keep_largest_area(t, 0, 1)
        torch.cuda.synchronize()
        best_dev = min(best_dev, (time.perf_counter() - t0) / 50)
        t0 = time.perf_counter()
        for _ in range(20):
            sp.label_image(img, 0)
        best_abi = min(best_abi, (time.perf_counter() - t0) / 20)
    assert out.shape == img.shape
    assert best_dev < 100e-6, "keep_largest_area on the device: %.1f us" % (best_dev * 1e6)
    assert best_abi < 400e-6, "label_image per host image: %.1f us" % (best_abi * 1e6)


def test_batches_of_images_on_the_device(dev, oracle):
    """every image of a batch labelled on its own, five launches for all of them: each against the oracle; tables with less room than an
    image has components keep their first entries and the count says how many there are"""
    import torch

    rng = np.random.default_rng(21)
    for dt, (n, h, w) in ((np.uint16, (9, 67, 131)), (np.float32, (5, 40, 64)), (np.uint8, (33, 20, 70)), (np.int64, (3, 128, 257))):
        frames = np.stack([np.kron(rng.integers(0, 4, (h // 4 + 1, w // 4 + 1)), np.ones((4, 4), dtype=np.int64))[:h, :w] for _ in range(n)])
        frames[1] = 0  # an image without a component
        frames[2] = 3  # one flat component
        frames = frames.astype(dt)
        if np.dtype(dt).kind == "f":
            frames[rng.random(frames.shape) < 0.01] = np.nan
        t = torch.from_numpy(frames).cuda()
        lab, area, xy, count = dev.label_images(t, 0, table_entries=h * w + 1)
        keep = dev.keep_largest_areas(t, 0, 6).cpu().numpy()
        lab, area, xy, count = lab.cpu().numpy(), area.cpu().numpy(), xy.cpu().numpy(), count.cpu().numpy()
        for i in range(n):
            exp = oracle.label_image(frames[i], 0)
            k = int(count[i])
            assert k == exp[1].size and np.array_equal(lab[i], exp[0]) and np.array_equal(area[i, :k], exp[1]) and np.array_equal(xy[i, :k], exp[2]), (dt, i)
            assert np.array_equal(keep[i], oracle.keep_largest_area(frames[i], 0, 6)), (dt, i)
        small = 4
        lab2, area2, xy2, count2 = dev.label_images(t, 0, table_entries=small)
        assert np.array_equal(lab2.cpu().numpy(), lab) and np.array_equal(count2.cpu().numpy(), count)
        for i in range(n):
            k = min(int(count[i]), small)
            assert np.array_equal(area2[i, :k].cpu().numpy(), area[i, :k]) and np.array_equal(xy2[i, :k].cpu().numpy(), xy[i, :k])
            assert not area2[i, k:].any()  # nothing written past what there is room for
