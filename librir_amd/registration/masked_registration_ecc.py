"""Sub-pixel translation tracking of a movie, same interface as librir's ``MaskedRegistratorECC``
(reference src/python/librir/registration/masked_registration_ecc.py:20-216).

The reference gets the alignment itself from OpenCV (``cv2.findTransformECC`` with MOTION_TRANSLATION,
``:166-168``); here that one call is ``find_transform_ecc_translation`` below, the ECC iterations running on
the MI355X (librir_amd/csrc/ecc_kernels.hip).  Everything around it - gaussian pre-filter, crop, optional
low-percentile mask, min-max normalisation, the confidence-driven change of reference image, the 4-column
TSV that ``load_motion_correction_file`` reads - follows the reference class step by step.
"""
import ctypes as ct

import numpy as np

from ..low_level.misc import _lib, last_error
from ..signal_processing.rir_signal_processing import find_median_pixel, gaussian_filter, translate

_lib.find_transform_ecc_translation.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double,
                                                ct.POINTER(ct.c_double)]


def find_transform_ecc_translation(template, image, warp_matrix=None, max_iterations=500, eps=1e-3, mask=None):
    """Counterpart of ``cv2.findTransformECC(template, image, warp_matrix, cv2.MOTION_TRANSLATION,
    (EPS | COUNT, max_iterations, eps), mask, 1)`` -> ``(cc, warp_matrix)`` with a 2x3 float32 matrix.
    Raises RuntimeError where OpenCV raises (no overlap, no convergence)."""
    t = np.ascontiguousarray(template, dtype=np.float32)
    im = np.ascontiguousarray(image, dtype=np.float32)
    if t.ndim != 2 or t.shape != im.shape:
        raise RuntimeError("find_transform_ecc_translation: template and image must be 2-D and of the same shape")
    m = None
    if mask is not None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape != t.shape:
            raise RuntimeError("find_transform_ecc_translation: mask shape")
    wm = np.eye(2, 3, dtype=np.float32) if warp_matrix is None else np.array(warp_matrix, dtype=np.float32)
    w2 = np.array([wm[0, 2], wm[1, 2]], dtype=np.float32)
    cc = ct.c_double(0)
    r = _lib.find_transform_ecc_translation(t.ctypes.data, im.ctypes.data, None if m is None else m.ctypes.data, t.shape[1], t.shape[0],
                                            w2.ctypes.data, int(max_iterations), float(eps), ct.byref(cc))
    if r != 0:
        raise RuntimeError("find_transform_ecc_translation: %s" % last_error())
    wm = np.eye(2, 3, dtype=np.float32)
    wm[0, 2], wm[1, 2] = w2[0], w2[1]
    return cc.value, wm


class MaskedRegistratorECC:
    """First image through ``start()``, the following ones through ``compute()``; the translations from the
    very first image accumulate in ``x`` / ``y``, the correlation coefficients in ``confidences``."""

    def __init__(self, window_factorh=0.7, window_factorv=0.7, sigma=0.5, mask=None, median=1, ref=None, pre_process=None, view=None):
        self.sigma = sigma
        self.x = []
        self.y = []
        self.confidences = []
        self.ref_img = None
        self.ref = ref
        if ref is not None and pre_process is not None:
            self.ref = pre_process(ref)
        if sigma > 0 and self.ref is not None:
            self.ref = gaussian_filter(self.ref, sigma)
        self.mask_ref_img = None
        self.window_factorH = window_factorh
        self.window_factorV = window_factorv
        shape = (512, 640)  # the crop window is defined on the camera format, like upstream (:78)
        self.subW = int(shape[1] * self.window_factorH)
        self.subH = int(shape[0] * self.window_factorV)
        self.startX = int((shape[1] - self.subW) / 2)
        self.startY = int((shape[0] - self.subH) / 2)
        self.mask = mask
        self.conf_thresh = None
        self.pre_process = pre_process
        self.view = view
        self.median = median
        self.start_mat = np.eye(2, 3, dtype=np.float32)
        self.number_of_iterations = 500  # :133
        self.termination_eps = 1e-3      # :137

    def _window(self, img):
        return img[self.startY:self.startY + self.subH, self.startX:self.startX + self.subW]

    def start(self, img):
        if self.pre_process is not None:
            img = self.pre_process(img)
        if self.sigma > 0:
            img = gaussian_filter(img, self.sigma)
        self.ref_img = self._window(img)
        if self.mask is not None:
            self.mask = self._window(self.mask)
        self.x.append(0)
        self.y.append(0)
        self.confidences.append(1)

    def compute(self, img):
        if self.pre_process is not None:
            img = self.pre_process(img)
        if self.sigma > 0:
            img = gaussian_filter(img, self.sigma)
        new_im = self._window(img).copy()
        im1 = np.array(self.ref_img if self.ref is None else self.ref, dtype=np.float32)
        im2 = np.array(new_im, dtype=np.float32)
        mask = self.mask
        if self.median < 1:  # dynamic mask: clip everything above the chosen percentile (:152-160)
            thresh = max(find_median_pixel(new_im, self.median, mask), find_median_pixel(self.ref_img, self.median, mask))
            sel = (im1 > thresh) | (im2 > thresh)
            im1[sel] = thresh
            im2[sel] = thresh
        mi, ma = np.min(im1), np.max(im1)
        im1 = (im1 - mi) / (ma - mi)
        mi, ma = np.min(im2), np.max(im2)
        im2 = (im2 - mi) / (ma - mi)
        cc, warp_matrix = find_transform_ecc_translation(im1, im2, self.start_mat, self.number_of_iterations, self.termination_eps, mask)
        self.start_mat = warp_matrix
        shift = [warp_matrix[1, 2], warp_matrix[0, 2]]
        self.confidences.append(cc)
        self.x.append(shift[1])
        self.y.append(shift[0])
        if len(self.confidences) > 20 and self.ref is None:  # change of reference image on a confidence drop (:177-189)
            if self.conf_thresh is None:
                self.conf_thresh = np.min(self.confidences) - 2 * np.std(self.confidences)
            if cc < self.conf_thresh:
                self.ref_img = translate(new_im, -shift[1], -shift[0])
                self.start_mat = np.eye(2, 3, dtype=np.float32)
        return shift

    def append_last_coordinates_and_confidence(self):
        self.x.append(self.x[-1])
        self.y.append(self.y[-1])
        self.confidences.append(self.confidences[-1])

    def return_coordinates_and_confidence_values(self):
        return np.array([self.x, self.y, self.confidences]).T

    @property
    def stabilisation_data(self):
        import pandas as pd

        return pd.DataFrame(data=self.return_coordinates_and_confidence_values(),
                            columns=["x-axis translations", "y-axis translations", "Confidence level"])

    def to_reg_file(self, dest_file):
        """Tab-separated, one header line, index + 3 columns: what ``load_motion_correction_file`` parses
        (IRFileLoader.cpp:822-847) and what pandas' ``to_csv(sep="\\t")`` writes upstream (:214-215)."""
        arr = self.return_coordinates_and_confidence_values()
        with open(dest_file, "w") as f:
            f.write("\tx-axis translations\ty-axis translations\tConfidence level\n")
            for i, (x, y, c) in enumerate(arr):
                f.write("%d\t%s\t%s\t%s\n" % (i, repr(float(x)), repr(float(y)), repr(float(c))))
