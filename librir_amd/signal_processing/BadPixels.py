"""Bad pixel correction object (reference src/python/librir/signal_processing/BadPixels.py:1-30)."""
import numpy as np

from .rir_signal_processing import bad_pixels_correct, bad_pixels_create, bad_pixels_destroy


class BadPixels(object):
    """Detects the bad pixels on a first image, then corrects any image of the same stream."""

    handle = 0

    def __init__(self, first_image):
        self.handle = bad_pixels_create(first_image)
        self.shape = tuple(np.asarray(first_image).shape)

    def __del__(self):
        if self.handle > 0:
            try:
                bad_pixels_destroy(self.handle)
            except Exception:
                pass
            self.handle = 0

    def correct(self, img):
        if tuple(np.asarray(img).shape) != self.shape:
            raise RuntimeError("BadPixels.correct: wrong image shape")
        return bad_pixels_correct(self.handle, img)
