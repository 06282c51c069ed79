"""librir_amd - MI355X-native implementation of librir's per-frame 16-bit IR hot path.

Python host code over a thin C-ABI HIP shared object (``libs/librir_amd.so``).  Sub-packages mirror
the reference wrapper (``video_io``, ``signal_processing``, ``tools``, ``registration``); ``device``
adds the device-resident batch API used by pipelines and ``bench.py``.
"""
__version__ = "0.1.0"

# The reference package imports its modules and main classes at the top level (src/python/librir/__init__.py:4-12: librir.IRMovie,
# librir.rir_video_io, ...).  The same names resolve here, on first use - importing the package alone loads nothing.
_LAZY = {
    "misc": ("librir_amd.low_level.misc", None),
    "rir_signal_processing": ("librir_amd.signal_processing.rir_signal_processing", None),
    "BadPixels": ("librir_amd.signal_processing.BadPixels", "BadPixels"),
    "rir_tools": ("librir_amd.tools.rir_tools", None),
    "rir_video_io": ("librir_amd.video_io.rir_video_io", None),
    "IRMovie": ("librir_amd.video_io.IRMovie", "IRMovie"),
    "IRSaver": ("librir_amd.video_io.IRSaver", "IRSaver"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        module, attr = _LAZY[name]
        value = importlib.import_module(module)
        if attr:
            value = getattr(value, attr)
        globals()[name] = value
        return value
    if name == "rir_geometry":
        raise AttributeError("librir_amd has no geometry module: the polygon library is the reference's own (INTEGRATION.md section 1)")
    raise AttributeError("module 'librir_amd' has no attribute %r" % name)
