"""Sub-pixel translation tracking of a movie, same interface as librir's ``MaskedRegistratorECC``
(reference src/python/librir/registration/masked_registration_ecc.py:20-216).

The reference gets the alignment itself from OpenCV (``cv2.findTransformECC`` with MOTION_TRANSLATION,
``:166-168``); here that one call is ``find_transform_ecc_translation`` below, the ECC iterations running on
the MI355X (librir_amd/csrc/ecc_kernels.hip).  Everything around it - gaussian pre-filter, crop, optional
low-percentile mask, min-max normalisation, the confidence-driven change of reference image, the 4-column
TSV that ``load_motion_correction_file`` reads - follows the reference class step by step.
"""
import ctypes as ct
import os

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
    """Tracks the translation of every image of a sequence with respect to the first one.

    ``start(first_image)`` once, then ``compute(image)`` per image; results accumulate in ``x``, ``y`` (pixels, with
    respect to the very first image) and ``confidences`` (correlation coefficients).  Constructor arguments as upstream:
    the window factors pick a centred sub-window, ``sigma`` pre-filters with a gaussian (0 = off), ``mask`` is a static
    0/1 image, ``median`` < 1 keeps only that fraction of the lowest pixels (dynamic mask), ``ref`` fixes the reference
    image, ``pre_process`` is applied to every image first."""

    CAMERA_SHAPE = (512, 640)  # upstream defines the window on the camera format, whatever the image size (:78)
    HISTORY_BEFORE_RESET = 20  # confidences collected before a drop may trigger a change of reference (:177)

    def __init__(self, window_factorh=0.7, window_factorv=0.7, sigma=0.5, mask=None, median=1, ref=None, pre_process=None, view=None):
        self.sigma, self.mask, self.median, self.pre_process, self.view = sigma, mask, median, pre_process, view
        self.window_factorH, self.window_factorV = window_factorh, window_factorv
        rows, columns = self.CAMERA_SHAPE
        self.subW, self.subH = int(columns * window_factorh), int(rows * window_factorv)
        self.startX, self.startY = int((columns - self.subW) / 2), int((rows - self.subH) / 2)
        self.x, self.y, self.confidences = [], [], []
        self._dev = None          # the one-call-per-image path (start()): every step on the device, one upload per image
        self._ref_img = None      # registration window of the current reference image (``ref_img``)
        self.mask_ref_img = None
        self.conf_thresh = None   # fixed the first time HISTORY_BEFORE_RESET is exceeded: min - 2 std of the history
        self.start_mat = np.eye(2, 3, dtype=np.float32)
        self.number_of_iterations = 500   # criteria of the upstream call (:133-137)
        self.termination_eps = 1e-3
        self.ref = None
        if ref is not None:
            self.ref = self._prepared(ref)

    # ---- pieces ------------------------------------------------------------------------------------------------------
    def _prepared(self, img):
        """user pre-processing, then the gaussian pre-filter"""
        if self.pre_process is not None:
            img = self.pre_process(img)
        return gaussian_filter(img, self.sigma) if self.sigma > 0 else img

    def _window(self, img):
        return img[self.startY:self.startY + self.subH, self.startX:self.startX + self.subW]

    @staticmethod
    def _unit_range(im):
        lowest, highest = np.min(im), np.max(im)
        return (im - lowest) / (highest - lowest)

    def _record(self, tx, ty, confidence):
        self.x.append(tx)
        self.y.append(ty)
        self.confidences.append(confidence)

    # ---- the plain configuration in one library call per image ------------------------------------------------------------------
    # Without masks, percentile clipping, a fixed reference or user pre-processing - what upstream's own tests use - an image goes up
    # once and gaussian pre-filter, window normalisation and alignment are ONE call into the library (rir_ecc_register_frame_device,
    # through DeviceRegistratorECC: the same kernels in the same order as the calls below, the same track -
    # tests/test_gpu_registration.py), instead of five host round trips.  RIR_REGISTRATION_STEP_BY_STEP=1 keeps the calls below.
    def _one_call_path(self, img):
        if os.environ.get("RIR_REGISTRATION_STEP_BY_STEP") or self.pre_process is not None or self.mask is not None or self.median < 1 or self.ref is not None:
            return False
        if not isinstance(img, np.ndarray) or img.ndim != 2 or img.dtype not in (np.uint16, np.float32):
            return False
        if self.subW < 2 or self.subH < 2 or self.startX + self.subW > img.shape[1] or self.startY + self.subH > img.shape[0]:
            return False  # (numpy truncates such a window; the device entry point refuses it)
        try:
            import torch

            return bool(torch.cuda.is_available())
        except Exception:
            return False

    def _up(self, img, keep=False):
        """The image in device memory.  Through a page-locked staging tensor of this object (the copy into it cut over the library's
        helper threads, rir_host_copy), then one asynchronous transfer: a pageable array handed to torch's ``.cuda()`` is staged by
        the runtime at a fraction of the link's rate (100 us for a 640x512 float image, against 35).  ``keep``: the caller keeps the
        tensor (the first image may become the reference as it is): a tensor of its own; otherwise this object's buffer, overwritten
        by the next image."""
        import torch

        a = np.ascontiguousarray(img)
        key = (a.shape, a.dtype)
        if getattr(self, "_stage_key", None) != key:
            t = torch.from_numpy(np.empty(a.shape, a.dtype))  # (a dtype torch can hold: uint16 / float32, _one_call_path)
            self._stage = torch.empty(a.shape, dtype=t.dtype, pin_memory=True)
            self._stage_dev = torch.empty(a.shape, dtype=t.dtype, device="cuda")
            self._stage_key = key
            _lib.rir_host_copy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int64]
        _lib.rir_host_copy(self._stage.data_ptr(), a.ctypes.data, a.nbytes)
        if keep:
            return self._stage.to("cuda", non_blocking=False)
        self._stage_dev.copy_(self._stage, non_blocking=True)  # (the staging tensor is written again by the next call, after this call's results)
        return self._stage_dev

    @property
    def ref_img(self):
        if self._dev is not None and self._ref_img is None:
            self._ref_img = self._dev.reference_window().cpu().numpy()
        return self._ref_img

    @ref_img.setter
    def ref_img(self, value):
        self._ref_img = value

    def _leave_one_call_path(self):
        """the state of the device-side registrator back into this object (an image the one-call path does not take came along)"""
        dev, self._ref_img = self._dev, None
        ref = self.ref_img
        self._dev = None
        self.ref_img = ref
        self.start_mat = np.eye(2, 3, dtype=np.float32)
        self.start_mat[0, 2], self.start_mat[1, 2] = dev.warp[0], dev.warp[1]
        self.conf_thresh = dev.conf_thresh

    # ---- the two calls ---------------------------------------------------------------------------------------------------
    def start(self, img):
        if self._one_call_path(img):
            from .device_registration import DeviceRegistratorECC

            dev = DeviceRegistratorECC(self.window_factorH, self.window_factorV, self.sigma, shape=self.CAMERA_SHAPE)
            dev.subW, dev.subH, dev.startX, dev.startY = self.subW, self.subH, self.startX, self.startY  # (as they are NOW: callers do adjust them)
            dev.x, dev.y, dev.confidences = self.x, self.y, self.confidences  # (one set of lists)
            dev.start(self._up(img, keep=True))
            self._dev, self._ref_img, self._shape = dev, None, (img.shape, img.dtype)
            return
        self.ref_img = self._window(self._prepared(img))
        if self.mask is not None:
            self.mask = self._window(self.mask)
        self._record(0, 0, 1)

    def compute(self, img):
        if self._dev is not None:
            if isinstance(img, np.ndarray) and (img.shape, img.dtype) == self._shape and self._one_call_path(img):
                dev = self._dev
                dev.number_of_iterations, dev.termination_eps, dev.conf_thresh = self.number_of_iterations, self.termination_eps, self.conf_thresh
                reference = dev._ref_n
                dy, dx = dev.compute(self._up(img))
                self.x[-1], self.y[-1] = np.float32(dx), np.float32(dy)  # (the types the calls below leave)
                self.start_mat = np.eye(2, 3, dtype=np.float32)
                self.start_mat[0, 2], self.start_mat[1, 2] = dev.warp[0], dev.warp[1]
                self.conf_thresh = dev.conf_thresh
                if dev._ref_n is not reference:
                    self._ref_img = None  # (the reference changed: fetched when somebody asks for it)
                return [np.float32(dy), np.float32(dx)]
            self._leave_one_call_path()
        current = self._window(self._prepared(img)).copy()
        template = np.array(self.ref_img if self.ref is None else self.ref, dtype=np.float32)
        moving = np.array(current, dtype=np.float32)
        if self.median < 1:  # dynamic mask (:152-160): everything above the chosen percentile of either image is clipped
            level = max(find_median_pixel(current, self.median, self.mask), find_median_pixel(self.ref_img, self.median, self.mask))
            too_bright = (template > level) | (moving > level)
            template[too_bright] = level
            moving[too_bright] = level
        cc, warp = find_transform_ecc_translation(self._unit_range(template), self._unit_range(moving), self.start_mat,
                                                  self.number_of_iterations, self.termination_eps, self.mask)
        self.start_mat = warp  # the next image starts from this one's result
        shift = [warp[1, 2], warp[0, 2]]  # (dy, dx), the order upstream returns
        self._record(shift[1], shift[0], cc)
        if self.ref is None and len(self.confidences) > self.HISTORY_BEFORE_RESET:
            if self.conf_thresh is None:
                self.conf_thresh = np.min(self.confidences) - 2 * np.std(self.confidences)
            if cc < self.conf_thresh:  # the scene changed too much: this image, shifted back, becomes the reference (:179-189)
                self.ref_img = translate(current, -shift[1], -shift[0])
                self.start_mat = np.eye(2, 3, dtype=np.float32)
        return shift

    # ---- results -------------------------------------------------------------------------------------------------------------
    def append_last_coordinates_and_confidence(self):
        """Repeat the last result (an image that could not be registered keeps its predecessor's translation)."""
        self._record(self.x[-1], self.y[-1], self.confidences[-1])

    def return_coordinates_and_confidence_values(self):
        return np.array([self.x, self.y, self.confidences]).T

    @property
    def stabilisation_data(self):
        import pandas as pd

        return pd.DataFrame(data=self.return_coordinates_and_confidence_values(),
                            columns=["x-axis translations", "y-axis translations", "Confidence level"])

    def to_reg_file(self, dest_file):
        """Tab-separated text, one header line, then ``index x y confidence`` per image: the layout pandas'
        ``to_csv(sep="\t")`` gives upstream (:214-215) and ``load_motion_correction_file`` parses
        (IRFileLoader.cpp:822-847: columns 1 and 2 are x and y)."""
        rows = self.return_coordinates_and_confidence_values()
        with open(dest_file, "w") as out:
            out.write("\t".join(["", "x-axis translations", "y-axis translations", "Confidence level"]) + "\n")
            for index, (tx, ty, confidence) in enumerate(rows):
                out.write("\t".join([str(index), repr(float(tx)), repr(float(ty)), repr(float(confidence))]) + "\n")


def manage_computation_and_tries(img, regis_obj):
    """``regis_obj.compute(img)`` with up to five attempts (reference masked_registration_ecc.py:218-245): an alignment that does not
    converge is tried again with the percentile of the dynamic mask lowered by 0.01 each time; after five failures the image takes the
    translation and confidence of its predecessor.  The failure upstream catches is ``cv2.error``; here the alignment raises
    ``RuntimeError`` where OpenCV raises.  Returns ``regis_obj``."""
    attempts, limit = 0, 5
    while attempts < limit:
        try:
            regis_obj.compute(img)
            if regis_obj.median < 1:
                regis_obj.median = 1
            break
        except RuntimeError:
            regis_obj.median -= 0.01
            attempts += 1
            print("try number : {}".format(attempts))
    if attempts >= limit:
        regis_obj.append_last_coordinates_and_confidence()
        print("took previous estimates.")
    return regis_obj
