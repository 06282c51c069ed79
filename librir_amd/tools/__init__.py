"""Mirror of librir's ``tools`` Python package (reference src/python/librir/tools/)."""
from .FileAttributes import FileAttributes
from .rir_tools import zstd_compress, zstd_compress_bound, zstd_decompress, zstd_decompress_bound

__all__ = ["FileAttributes", "zstd_compress", "zstd_decompress", "zstd_compress_bound", "zstd_decompress_bound"]
