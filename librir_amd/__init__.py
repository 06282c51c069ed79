"""librir_amd - MI355X-native implementation of librir's per-frame 16-bit IR hot path.

Python host code over a thin C-ABI HIP shared object (``libs/librir_amd.so``).  Sub-packages mirror
the reference wrapper (``video_io``, ``signal_processing``, ``tools``, ``registration``); ``device``
adds the device-resident batch API used by pipelines and ``bench.py``.
"""
__version__ = "0.1.0"
