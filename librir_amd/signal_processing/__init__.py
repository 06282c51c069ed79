"""Mirror of librir's ``signal_processing`` Python package (reference
src/python/librir/signal_processing/__init__.py) over the HIP shared object."""
from .BadPixels import BadPixels
from .rir_signal_processing import (bad_pixels_correct, bad_pixels_create, bad_pixels_destroy, extract_times, filter_chain, find_median_pixel,
                                    gaussian_filter, keep_largest_area, label_image, resample_time_serie, translate)

__all__ = ["BadPixels", "translate", "gaussian_filter", "find_median_pixel", "bad_pixels_create", "bad_pixels_correct",
           "bad_pixels_destroy", "extract_times", "resample_time_serie", "label_image", "keep_largest_area", "filter_chain"]
