"""Device-resident batch API over the C ABI in ``include/rir_amd_device.h``.

Inputs and outputs are ``torch`` CUDA(HIP) tensors; torch is plumbing only (device memory and
streams) - every operation is one call into ``librir_amd.so`` on torch's current HIP stream.
"""
import ctypes as ct

import numpy as np
import torch

from .low_level.misc import _lib, last_error

DEFAULT_GOP = 50  # reference key-frame cadence, src/cpp/video_io/h264.cpp:1662-1665

_DTYPE_CHARS = {
    torch.bool: "?",
    torch.int8: "b",
    torch.uint8: "B",
    torch.int16: "h",
    torch.uint16: "H",
    torch.int32: "i",
    torch.uint32: "I",
    torch.int64: "l",
    torch.uint64: "L",
    torch.float32: "f",
    torch.float64: "d",
}
_NP_OF = {
    torch.bool: np.bool_,
    torch.int8: np.int8,
    torch.uint8: np.uint8,
    torch.int16: np.int16,
    torch.uint16: np.uint16,
    torch.int32: np.int32,
    torch.uint32: np.uint32,
    torch.int64: np.int64,
    torch.uint64: np.uint64,
    torch.float32: np.float32,
    torch.float64: np.float64,
}


class CodecLayout(ct.Structure):
    _fields_ = [
        ("width", ct.c_int),
        ("height", ct.c_int),
        ("nframes", ct.c_int),
        ("gop", ct.c_int),
        ("ntiles", ct.c_int),
        ("nchunks", ct.c_int),
        ("hdr_bytes", ct.c_int64),
        ("tile_off_bytes", ct.c_int64),
        ("chunk_off_bytes", ct.c_int64),
        ("stream_max_bytes", ct.c_int64),
        ("workspace_bytes", ct.c_int64),
    ]


class CodecSlots(ct.Structure):
    _fields_ = [("slots_offset_bytes", ct.c_int64), ("slot_words", ct.c_int64), ("seg_words_offset_bytes", ct.c_int64), ("nslots", ct.c_int64)]


class CodecPackedLayout(ct.Structure):
    _fields_ = [("ntiles", ct.c_int), ("nchunks", ct.c_int), ("hdr_bytes", ct.c_int64), ("seg_pos_bytes", ct.c_int64), ("seg_words_bytes", ct.c_int64),
                ("stream_budget_bytes", ct.c_int64), ("stream_max_bytes", ct.c_int64), ("workspace_min_bytes", ct.c_int64),
                ("workspace_max_bytes", ct.c_int64)]


_vp = ct.c_void_p
_lib.rir_codec_slots_query.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(CodecSlots)]
_lib.rir_device_available.restype = ct.c_int
_lib.rir_stream_synchronize.argtypes = [_vp]
_lib.rir_codec_layout_query.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(CodecLayout)]
_lib.rir_codec_encode_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, _vp, _vp, ct.c_longlong, _vp]
_lib.rir_codec_encode_single_pass_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, _vp, _vp, ct.c_longlong, _vp]
_lib.rir_codec_encode_status.argtypes = [_vp, _vp]
_lib.rir_codec_encode_tiles_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, ct.c_longlong, _vp]
_lib.rir_codec_encode_compact_device.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, _vp, ct.c_longlong, _vp]
_lib.rir_codec_workspace_create_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_longlong, ct.POINTER(ct.c_void_p),
                                                   ct.POINTER(ct.c_float), ct.POINTER(ct.c_int), _vp]
_lib.rir_buffer_create_beside_device.argtypes = [_vp, ct.c_longlong, ct.c_longlong, ct.c_int, ct.c_longlong, ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_float),
                                                 ct.POINTER(ct.c_int), _vp]
_lib.rir_buffer_destroy_device.argtypes = [_vp]
_lib.rir_buffer_destroy_device.restype = None
_lib.rir_codec_workspace_destroy_device.argtypes = [_vp]
_lib.rir_codec_workspace_destroy_device.restype = None
_lib.rir_codec_decode_slots_device.argtypes = [_vp, _vp, ct.c_longlong, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp]
_lib.rir_codec_packed_query.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(CodecPackedLayout)]
_lib.rir_codec_encode_packed_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, _vp, ct.c_longlong, _vp, ct.c_longlong, _vp]
_lib.rir_codec_encode_packed_launch_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, _vp, ct.c_longlong, _vp, ct.c_longlong, _vp]
_lib.rir_codec_packed_reset_device.argtypes = [_vp, ct.c_longlong, _vp]
_lib.rir_codec_encode_packed_status.argtypes = [_vp, ct.POINTER(ct.c_ulonglong), _vp]
_lib.rir_codec_decode_packed_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_longlong, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp]
_lib.rir_codec_decode_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_longlong, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp]
_lib.rir_codec_decode_chunks_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_longlong, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, ct.c_longlong, _vp,
                                                _vp, _vp]
_lib.rir_translate_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, ct.c_int, _vp, ct.c_char_p, _vp]
_lib.rir_gaussian_filter_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, ct.c_float, _vp]
_lib.rir_gaussian_filter_u16_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, ct.c_float, _vp]
_lib.rir_translate_f32_u16_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, ct.c_int, _vp, ct.c_char_p, _vp]
_lib.rir_filter_chain_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, ct.c_float, _vp, ct.c_int, _vp, ct.c_char_p, _vp]
_lib.rir_find_median_pixel_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_float, _vp, _vp, _vp]
_lib.rir_bad_pixels_create_device.argtypes = [_vp, ct.c_int, ct.c_int, _vp]
_lib.rir_bad_pixels_create_rows_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, _vp]
_lib.rir_bad_pixels_correct_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, _vp]
_lib.rir_bad_pixels_info.argtypes = [ct.c_int, _vp, _vp, ct.c_int]
_lib.rir_remove_bad_pixels_device.argtypes = [ct.c_int, _vp, ct.c_int, ct.c_int, _vp]
_lib.rir_remove_motion_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp]
_lib.rir_median_filter_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp]
_lib.bad_pixels_destroy.argtypes = [ct.c_int]
_lib.rir_label_workspace_bytes.argtypes = [ct.c_int, ct.c_int]
_lib.rir_label_workspace_bytes.restype = ct.c_size_t
_lib.rir_label_workspace_bytes_batch.argtypes = [ct.c_int, ct.c_int, ct.c_int]
_lib.rir_label_workspace_bytes_batch.restype = ct.c_size_t
_lib.rir_label_images_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, ct.c_int, _vp, _vp, ct.c_size_t, _vp]
_lib.rir_keep_largest_areas_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, ct.c_int, _vp, ct.c_size_t, _vp]
_lib.rir_label_image_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, ct.c_int, _vp, _vp, _vp, _vp, _vp, ct.c_size_t, _vp]
_lib.rir_keep_largest_area_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, ct.c_int, _vp, ct.c_int, _vp, ct.c_size_t, _vp]
_lib.rir_lossy_create.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_double, ct.c_int, ct.c_int, ct.c_int]
_lib.rir_lossy_step_device.argtypes = [ct.c_int, _vp, _vp, ct.c_int, ct.c_int, _vp, _vp, _vp]
_lib.rir_lossy_step_multi_device.argtypes = [_vp, ct.c_int, _vp, _vp, ct.c_int, ct.c_int, _vp, _vp, _vp]
_lib.rir_lossy_destroy.argtypes = [ct.c_int]
_lib.rir_lossy_status.argtypes = [ct.c_int, _vp]
_lib.rir_lossy_path_stats.argtypes = [ct.c_int, ct.POINTER(ct.c_int), _vp]
_lib.rir_lossy_spec_stats.argtypes = [ct.c_int, ct.POINTER(ct.c_int), _vp]
_lib.rir_lossy_set_errors.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_double]
_lib.rir_split_planes_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, _vp]
_lib.rir_merge_planes_device.argtypes = [_vp, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp]
_lib.bad_pixels_destroy.restype = None


def device_available():
    return bool(_lib.rir_device_available())


def _stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(r, what):
    if r != 0:
        raise RuntimeError("%s failed: %s" % (what, last_error()))


def _frames3(t, dtype=None):
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3 or not t.is_cuda:
        raise RuntimeError("expected a CUDA tensor of shape (n, h, w)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("expected dtype %s" % dtype)
    return t.contiguous()


def codec_layout(width, height, nframes, gop=DEFAULT_GOP):
    L = CodecLayout()
    _check(_lib.rir_codec_layout_query(width, height, nframes, gop, ct.byref(L)), "rir_codec_layout_query")
    return L


class EncodedBatch:
    """Device-resident compressed batch (tables + compact stream), format RIRB1."""

    def __init__(self, layout, hdr, tile_off, chunk_off, stream):
        self.layout = layout
        self.hdr = hdr  # int64(bit pattern uint64) [nchunks, ntiles, gop]
        self.tile_off = tile_off  # int32(bit pattern uint32) [nchunks, ntiles+1]
        self.chunk_off = chunk_off  # int64 [nchunks+1]
        self.stream = stream  # int64 words (capacity = worst case)

    def total_words(self):
        return int(self.chunk_off[-1].item())

    def compressed_bytes(self):
        """stream + tables: what a container has to store"""
        L = self.layout
        return self.total_words() * 8 + L.hdr_bytes + L.tile_off_bytes + L.chunk_off_bytes


class _LibraryBuffer:
    """Device memory allocated by the library (rir_codec_workspace_create_device), seen from torch as a uint8 tensor that keeps
    this object - and so the allocation - alive."""

    def __init__(self, ptr, nbytes, destroy=None):
        self.ptr, self.nbytes = int(ptr), int(nbytes)
        self._destroy = destroy if destroy is not None else _lib.rir_codec_workspace_destroy_device
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def tensor(self, device):
        return torch.as_tensor(self, device=device)

    def __del__(self):
        try:
            if self.ptr:
                self._destroy(ct.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass


def empty_beside(other, shape, dtype, tries=6, spacing_bytes=6 << 30):
    """A device tensor of ``shape`` / ``dtype`` in another placement class than the tensor ``other`` (rir_buffer_create_beside_device): the
    output buffer for a kernel that reads ``other`` and writes an output of similar size at the same pace - translate, gaussian_filter,
    filter_chain, median_filter are 5-10 % faster then (placement classes).  One-off set-up: candidates are
    allocated by the library, a streaming copy is timed on each, the rest is freed.  Returns (tensor, measured times in us, kept first)."""
    nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    nbytes = (nbytes + 15) // 16 * 16
    ptr = ct.c_void_p()
    times = (ct.c_float * (int(tries) + 1))()
    nt = ct.c_int(0)
    o = other.contiguous()
    _check(_lib.rir_buffer_create_beside_device(o.data_ptr(), o.numel() * o.element_size(), nbytes, int(tries), int(spacing_bytes), ct.byref(ptr), times,
                                                ct.byref(nt), _stream()), "rir_buffer_create_beside_device")
    owner = _LibraryBuffer(ptr.value, nbytes, _lib.rir_buffer_destroy_device)
    t = owner.tensor(other.device)[:int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()].view(dtype).view(*shape)
    t._rir_owner = owner  # (belt and braces: the allocation lives as long as the tensor object the caller holds)
    return t, [float(times[i]) for i in range(nt.value)]


class CodecContext:
    """Pre-allocated buffers for repeated encode/decode of one batch geometry (no allocation in
    the timed path)."""

    def __init__(self, width, height, nframes, gop=DEFAULT_GOP, device="cuda"):
        self.layout = L = codec_layout(width, height, nframes, gop)
        dev = torch.device(device)
        self.hdr = torch.zeros((L.nchunks, L.ntiles, L.gop), dtype=torch.int64, device=dev)
        self.tile_off = torch.zeros((L.nchunks, L.ntiles + 1), dtype=torch.int32, device=dev)
        self.chunk_off = torch.zeros((L.nchunks + 1,), dtype=torch.int64, device=dev)
        self.stream = torch.empty((L.stream_max_bytes // 8,), dtype=torch.int64, device=dev)
        self.workspace = torch.empty((L.workspace_bytes,), dtype=torch.uint8, device=dev)
        self.error = torch.zeros((1,), dtype=torch.int32, device=dev)

    def encode(self, frames, single_pass=False):
        """single_pass: the one-kernel encoder (dense stream written directly, decoupled look-back) - same outputs"""
        L = self.layout
        fr = _frames3(frames, torch.uint16)
        if tuple(fr.shape) != (L.nframes, L.height, L.width):
            raise RuntimeError("encode: frames do not match the context geometry")
        _check(
            (_lib.rir_codec_encode_single_pass_device if single_pass else _lib.rir_codec_encode_device)(
                fr.data_ptr(), L.width, L.height, L.nframes, L.gop, self.hdr.data_ptr(), self.tile_off.data_ptr(),
                self.chunk_off.data_ptr(), self.stream.data_ptr(), self.workspace.data_ptr(), L.workspace_bytes, _stream(),
            ),
            "rir_codec_encode_device",
        )
        return EncodedBatch(L, self.hdr, self.tile_off, self.chunk_off, self.stream)

    def place_workspace(self, frames, tries=6, spacing_bytes=6 << 30):
        """Replaces the encode workspace by one the LIBRARY allocates where the packing kernel runs fast for THIS frames buffer
        (rir_codec_workspace_create_device; DESIGN.md §7: device allocations fall into a few placement classes, and the kernel -
        frames in, slots out at the same pace - is 10 % slower when both are of one class).  One-off set-up of a few
        milliseconds; the candidates that lose and the spacers between them are freed by the library before it returns, torch's
        allocator is not touched.  There are three classes and they come in runs of 8-24 GiB of the address space
        (profiles/r03_placement/): candidates ``spacing_bytes`` apart, at most ``tries`` of them.  Returns the measured packing
        times in microseconds, the kept candidate's first."""
        L = self.layout
        fr = _frames3(frames, torch.uint16)
        if tuple(fr.shape) != (L.nframes, L.height, L.width):
            raise RuntimeError("place_workspace: frames do not match the context geometry")
        ptr = ct.c_void_p()
        times = (ct.c_float * (int(tries) + 1))()
        nt = ct.c_int(0)
        _check(_lib.rir_codec_workspace_create_device(fr.data_ptr(), L.width, L.height, L.nframes, L.gop, int(tries), int(spacing_bytes), ct.byref(ptr),
                                                      times, ct.byref(nt), _stream()), "rir_codec_workspace_create_device")
        owner = _LibraryBuffer(ptr.value, L.workspace_bytes)
        self.workspace = owner.tensor(self.hdr.device)
        self._workspace_owner = owner  # (the allocation lives as long as this context uses it, whatever torch keeps alive)
        return [float(times[i]) for i in range(nt.value)]

    def encode_status(self):
        """0 when the last single-pass encode completed, 1 when one of its look-backs gave up (waits for the stream)"""
        return int(_lib.rir_codec_encode_status(self.workspace.data_ptr(), _stream()))

    def encode_tiles(self, frames):
        """stage 1 only (single pass over the raw frames); finish with encode_compact()"""
        L = self.layout
        fr = _frames3(frames, torch.uint16)
        if tuple(fr.shape) != (L.nframes, L.height, L.width):
            raise RuntimeError("encode: frames do not match the context geometry")
        _check(_lib.rir_codec_encode_tiles_device(fr.data_ptr(), L.width, L.height, L.nframes, L.gop, self.hdr.data_ptr(),
                                                  self.workspace.data_ptr(), L.workspace_bytes, _stream()), "rir_codec_encode_tiles_device")

    def slots(self):
        """Views of the slotted form inside the workspace (valid after encode_tiles): (seg_words int32[nchunks, ntiles] - bit
        pattern uint32 -, slots int64[nchunks, ntiles, slot_words] - bit pattern uint64)."""
        L = self.layout
        S = CodecSlots()
        _check(_lib.rir_codec_slots_query(L.width, L.height, L.nframes, L.gop, ct.byref(S)), "rir_codec_slots_query")
        n = L.nchunks * L.ntiles
        seg = self.workspace[S.seg_words_offset_bytes:S.seg_words_offset_bytes + n * 4].view(torch.int32).view(L.nchunks, L.ntiles)
        sl = self.workspace[S.slots_offset_bytes:S.slots_offset_bytes + n * S.slot_words * 8].view(torch.int64).view(L.nchunks, L.ntiles, S.slot_words)
        return seg, sl

    def slots_payload_bytes(self):
        """payload bytes of the slotted batch in the workspace (sum of the segment lengths)"""
        return int(self.slots()[0].to(torch.int64).sum().item()) * 8

    def decode_slots(self, out=None, check=True):
        """decode of the slotted form left by encode_tiles (rir_codec_decode_slots_device): no second encoder pass"""
        L = self.layout
        if out is None:
            out = torch.empty((L.nframes, L.height, L.width), dtype=torch.uint16, device=self.hdr.device)
        if check:
            self.error.zero_()
        _check(_lib.rir_codec_decode_slots_device(self.hdr.data_ptr(), self.workspace.data_ptr(), L.workspace_bytes, L.width, L.height, L.nframes,
                                                  L.gop, out.data_ptr(), self.error.data_ptr(), _stream()), "rir_codec_decode_slots_device")
        if check and int(self.error.item()) != 0:
            raise RuntimeError("rir_codec_decode_slots_device: malformed stream")
        return out

    def encode_compact(self):
        L = self.layout
        _check(_lib.rir_codec_encode_compact_device(L.width, L.height, L.nframes, L.gop, self.tile_off.data_ptr(), self.chunk_off.data_ptr(),
                                                    self.stream.data_ptr(), self.workspace.data_ptr(), L.workspace_bytes, _stream()),
               "rir_codec_encode_compact_device")
        return EncodedBatch(L, self.hdr, self.tile_off, self.chunk_off, self.stream)

    def decode(self, enc, out=None, check=True):
        L = self.layout
        if out is None:
            out = torch.empty((L.nframes, L.height, L.width), dtype=torch.uint16, device=self.hdr.device)
        if check:
            self.error.zero_()
        _check(
            _lib.rir_codec_decode_device(
                enc.hdr.data_ptr(), enc.tile_off.data_ptr(), enc.chunk_off.data_ptr(), enc.stream.data_ptr(), enc.stream.numel(), L.width, L.height,
                L.nframes, L.gop, out.data_ptr(), self.error.data_ptr(), _stream(),
            ),
            "rir_codec_decode_device",
        )
        if check and int(self.error.item()) != 0:
            raise RuntimeError("rir_codec_decode_device: malformed stream")
        return out


class PackedBatch:
    """An encoded batch in the packed form: record headers, one (position, length) pair per (chunk, tile) segment and a stream
    buffer whose two ends hold the payload without holes - ``[0, low)`` and ``[capacity - high, capacity)`` words.  ``nbytes()``
    is what it occupies and what has to be kept, sent or written."""

    def __init__(self, codec, hdr, seg_pos, seg_words, stream, low, high):
        self.codec, self.hdr, self.seg_pos, self.seg_words, self.stream, self.low, self.high = codec, hdr, seg_pos, seg_words, stream, int(low), int(high)
        self.words = self.low + self.high

    def extents(self):
        """the two pieces of the payload as views of the stream buffer"""
        return self.stream[:self.low], self.stream[self.stream.numel() - self.high:]

    def payload_bytes(self):
        return self.words * 8

    def nbytes(self):
        return self.words * 8 + self.hdr.numel() * 8 + self.seg_pos.numel() * 8 + self.seg_words.numel() * 4


class PackedCodec:
    """Encode / decode of one batch geometry through the PACKED form (rir_codec_encode_packed_device): one pass over the
    frames, the encoded batch = exactly its payload + tables.  ``stream_bytes``: capacity of the stream buffer - default the
    8 bit-per-pixel budget (half the raw size; the reference documents a factor of about 5, docs/video_io.md:13), "max" = room
    for any data.  ``workspace_bytes``: default the minimum (control block + a small arena), "max" = room for any data.  A batch
    that does not fit raises in ``finish()`` / ``encode(..., check=True)`` with the sizes it needs; ``grow()`` provides them."""

    def __init__(self, width, height, nframes, gop=DEFAULT_GOP, device="cuda", stream_bytes=None, workspace_bytes=None):
        self.width, self.height, self.nframes, self.gop = int(width), int(height), int(nframes), int(gop)
        self.P = P = CodecPackedLayout()
        _check(_lib.rir_codec_packed_query(width, height, nframes, gop, ct.byref(P)), "rir_codec_packed_query")
        self.device = dev = torch.device(device)
        self.hdr = torch.zeros((P.nchunks, P.ntiles, self.gop), dtype=torch.int64, device=dev)
        self.seg_pos = torch.zeros((P.nchunks, P.ntiles), dtype=torch.int64, device=dev)
        self.seg_words = torch.zeros((P.nchunks, P.ntiles), dtype=torch.int32, device=dev)
        sb = P.stream_budget_bytes if stream_bytes is None else (P.stream_max_bytes if stream_bytes == "max" else int(stream_bytes))
        wb = P.workspace_min_bytes if workspace_bytes is None else (P.workspace_max_bytes if workspace_bytes == "max" else int(workspace_bytes))
        self.stream = torch.empty((max(sb // 8, 1),), dtype=torch.int64, device=dev)
        self.workspace = torch.empty((max(wb, 4096),), dtype=torch.uint8, device=dev)
        self.error = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.raw_bytes = self.width * self.height * self.nframes * 2

    def reset(self):
        """the first half of encode(): zero the workspace's control block (a fill launch of its own)"""
        _check(_lib.rir_codec_packed_reset_device(self.workspace.data_ptr(), self.workspace.numel(), _stream()), "rir_codec_packed_reset_device")

    def encode(self, frames, check=False, reset=True):
        """Asynchronous; ``check=True`` (or a later ``finish()``) waits and returns the PackedBatch.  ``reset=False``: the caller has just
        called reset() on the same stream (the packing kernel alone, for timings)."""
        fr = _frames3(frames, torch.uint16)
        if tuple(fr.shape) != (self.nframes, self.height, self.width):
            raise RuntimeError("encode: frames do not match the codec geometry")
        _check((_lib.rir_codec_encode_packed_device if reset else _lib.rir_codec_encode_packed_launch_device)(fr.data_ptr(), self.width, self.height, self.nframes, self.gop, self.hdr.data_ptr(), self.seg_pos.data_ptr(),
                                                   self.seg_words.data_ptr(), self.stream.data_ptr(), self.stream.numel(), self.workspace.data_ptr(),
                                                   self.workspace.numel(), _stream()), "rir_codec_encode_packed_device")
        return self.finish() if check else None

    def status(self):
        """(code, low words, high words, arena words asked for): code 0 complete, bit 0 stream capacity, bit 1 arena exceeded; waits."""
        out = (ct.c_ulonglong * 3)()
        r = int(_lib.rir_codec_encode_packed_status(self.workspace.data_ptr(), out, _stream()))
        if r < 0:
            raise RuntimeError("rir_codec_encode_packed_status failed: %s" % last_error())
        return r, int(out[0]), int(out[1]), int(out[2])

    def finish(self):
        r, low, high, arena = self.status()
        if r != 0:
            raise RuntimeError("packed encode: the batch does not fit (code %d): it needs %d stream bytes (capacity %d) and asked for %d arena bytes "
                               "(capacity %d)" % (r, (low + high) * 8, self.stream.numel() * 8, arena * 8, self.workspace.numel() - 4096))
        return PackedBatch(self, self.hdr, self.seg_pos, self.seg_words, self.stream, low, high)

    def grow(self):
        """room for any data (after a batch did not fit)"""
        self.stream = torch.empty((self.P.stream_max_bytes // 8,), dtype=torch.int64, device=self.device)
        self.workspace = torch.empty((self.P.workspace_max_bytes,), dtype=torch.uint8, device=self.device)

    def decode(self, batch=None, out=None, check=True):
        hdr, pos, seg, st = (batch.hdr, batch.seg_pos, batch.seg_words, batch.stream) if batch is not None else (self.hdr, self.seg_pos, self.seg_words, self.stream)
        if out is None:
            out = torch.empty((self.nframes, self.height, self.width), dtype=torch.uint16, device=self.device)
        if check:
            self.error.zero_()
        _check(_lib.rir_codec_decode_packed_device(hdr.data_ptr(), pos.data_ptr(), seg.data_ptr(), st.data_ptr(), st.numel(), self.width, self.height,
                                                   self.nframes, self.gop, out.data_ptr(), self.error.data_ptr(), _stream()), "rir_codec_decode_packed_device")
        if check and int(self.error.item()) != 0:
            raise RuntimeError("rir_codec_decode_packed_device: malformed batch")
        return out


def decode_chunks(hdr, tile_off, chunk_off, stream, chunk_frames, out, gop, error):
    """Chunks that do not form one contiguous batch (gathered from several shards): table entry k = one chunk,
    ``chunk_frames[k] = (first frame, frame count)`` inside ``out`` (N, H, W) uint16.  ``error``: int32[1] device tensor,
    raised to 1 on malformed tables; asynchronous on the current stream (rir_codec_decode_chunks_device)."""
    n, h, w = out.shape
    nchunks = chunk_frames.shape[0]
    if hdr.shape[0] < nchunks or tile_off.shape[0] < nchunks or chunk_off.numel() < nchunks + 1 or chunk_frames.dtype != torch.int64:
        raise RuntimeError("decode_chunks: tables do not cover the chunks")
    _check(
        _lib.rir_codec_decode_chunks_device(hdr.data_ptr(), tile_off.data_ptr(), chunk_off.data_ptr(), stream.data_ptr(), stream.numel(), w, h,
                                            nchunks, gop, chunk_frames.data_ptr(), n, out.data_ptr(), error.data_ptr(), _stream()),
        "rir_codec_decode_chunks_device",
    )
    return out


def translate(frames, offsets, strategy="", background=0):
    """frames (n,h,w) any supported dtype; offsets: (dx,dy) or tensor (n,2) of float32 per-frame shifts."""
    fr = _frames3(frames)
    n, h, w = fr.shape
    ch = _DTYPE_CHARS.get(fr.dtype)
    if ch is None:
        raise RuntimeError("translate: unsupported dtype")
    off = torch.as_tensor(offsets, dtype=torch.float32, device=fr.device).contiguous()
    per_frame = 1 if off.dim() == 2 else 0
    if per_frame and off.shape[0] != n:
        raise RuntimeError("translate: one (dx,dy) pair per frame expected")
    # "noborder" leaves the pixels without a source as they are and the wrapper pre-fills with the input
    # (reference rir_signal_processing.py:54-55): "noborder_source" is that, without the copy
    dst = torch.empty_like(fr)
    if strategy in ("", "noborder"):
        strategy = "noborder_source"
    back = np.zeros(1, dtype=_NP_OF[fr.dtype])
    back[0] = background
    if strategy == "constant":
        strategy = "background"
    _check(
        _lib.rir_translate_device(ord(ch), fr.data_ptr(), dst.data_ptr(), w, h, n, off.data_ptr(), per_frame, back.ctypes.data,
                                  strategy.encode(), _stream()),
        "rir_translate_device",
    )
    return dst


def gaussian_filter(frames, sigma):
    """float32 frames, or uint16 frames (converted on the fly: same result as frames.float(), sigma < 2.5)."""
    if frames.dtype == torch.uint16:
        fr = _frames3(frames, torch.uint16)
        n, h, w = fr.shape
        dst = torch.empty((n, h, w), dtype=torch.float32, device=fr.device)
        _check(_lib.rir_gaussian_filter_u16_device(fr.data_ptr(), dst.data_ptr(), w, h, n, float(sigma), _stream()), "rir_gaussian_filter_u16_device")
        return dst
    fr = _frames3(frames, torch.float32)
    n, h, w = fr.shape
    dst = torch.empty_like(fr)
    _check(_lib.rir_gaussian_filter_device(fr.data_ptr(), dst.data_ptr(), w, h, n, float(sigma), _stream()), "rir_gaussian_filter_device")
    return dst


def translate_to_u16(frames, offsets, strategy="nearest", background=0):
    """translate(float32 frames).to(uint16) in one pass (strategies that write every pixel)."""
    fr = _frames3(frames, torch.float32)
    n, h, w = fr.shape
    if strategy in ("", "noborder"):
        raise RuntimeError("translate_to_u16: 'noborder' needs a pre-filled destination, use translate()")
    off = torch.as_tensor(offsets, dtype=torch.float32, device=fr.device).contiguous()
    per_frame = 1 if off.dim() == 2 else 0
    if per_frame and off.shape[0] != n:
        raise RuntimeError("translate: one (dx,dy) pair per frame expected")
    dst = torch.empty((n, h, w), dtype=torch.uint16, device=fr.device)
    back = np.array([background], dtype=np.uint16)
    if strategy == "constant":
        strategy = "background"
    _check(_lib.rir_translate_f32_u16_device(fr.data_ptr(), dst.data_ptr(), w, h, n, off.data_ptr(), per_frame, back.ctypes.data,
                                             strategy.encode(), _stream()), "rir_translate_f32_u16_device")
    return dst


def filter_chain(frames, bad_pixels, sigma, offsets, strategy="nearest", background=0, out=None):
    """``bad_pixels.correct`` -> ``gaussian_filter(sigma)`` -> ``translate(offsets, strategy)`` -> uint16, in ONE pass over the
    uint16 frames (4 bytes of HBM traffic per pixel instead of 14); bit-identical to the three calls.  ``bad_pixels``: a
    ``BadPixels`` object or None.  Strategies "nearest" and "background"; for the others run the three calls."""
    fr = _frames3(frames, torch.uint16)
    n, h, w = fr.shape
    if strategy == "constant":
        strategy = "background"
    if strategy not in ("nearest", "background"):
        raise RuntimeError("filter_chain: strategy must be 'nearest' or 'background'")
    off = torch.as_tensor(offsets, dtype=torch.float32, device=fr.device).contiguous()
    per_frame = 1 if off.dim() == 2 else 0
    if per_frame and off.shape[0] != n:
        raise RuntimeError("filter_chain: one (dx,dy) pair per frame expected")
    if out is not None and (tuple(out.shape) != (n, h, w) or out.dtype != torch.uint16 or not out.is_contiguous() or out.data_ptr() == fr.data_ptr()):
        raise RuntimeError("filter_chain: out must be a contiguous uint16 tensor of the frames' shape, not the input")
    dst = out if out is not None else torch.empty((n, h, w), dtype=torch.uint16, device=fr.device)  # (out: e.g. from empty_beside(frames, ...))
    back = np.array([background], dtype=np.uint16)
    handle = bad_pixels.handle if bad_pixels is not None else 0
    _check(_lib.rir_filter_chain_device(handle, fr.data_ptr(), dst.data_ptr(), w, h, n, float(sigma), off.data_ptr(), per_frame, back.ctypes.data,
                                        strategy.encode(), _stream()), "rir_filter_chain_device")
    return dst


def find_median_pixel(frames, percent=0.5, mask=None):
    fr = _frames3(frames, torch.uint16)
    n, h, w = fr.shape
    res = torch.zeros((n,), dtype=torch.int32, device=fr.device)
    hist = torch.empty((n, 65536), dtype=torch.int32, device=fr.device)
    mptr = None
    if mask is not None:
        mask = _frames3(mask, torch.uint8)
        mptr = mask.data_ptr()
    _check(_lib.rir_find_median_pixel_device(fr.data_ptr(), mptr, h * w, n, float(percent), res.data_ptr(), hist.data_ptr(), _stream()),
           "rir_find_median_pixel_device")
    return res


class BadPixels:
    """Device-side counterpart of librir's BadPixels (reference src/python/librir/signal_processing/BadPixels.py)."""

    def __init__(self, first_image, rows=None):
        img = _frames3(first_image, torch.uint16)
        _, h, w = img.shape
        if rows is None:
            self.handle = _lib.rir_bad_pixels_create_device(img.data_ptr(), w, h, _stream())
        else:
            self.handle = _lib.rir_bad_pixels_create_rows_device(img.data_ptr(), w, h, int(rows), _stream())
        if self.handle <= 0:
            raise RuntimeError("bad_pixels_create failed: %s" % last_error())
        self.shape = (h, w)
        info = (ct.c_int * 3)()
        _lib.rir_bad_pixels_info(self.handle, info, None, 0)
        self.count, self.floor_correct, self.floor_detect = info[0], info[1], info[2]

    def positions(self):
        xy = np.zeros((max(self.count, 1), 2), dtype=np.int32)
        info = (ct.c_int * 3)()
        _lib.rir_bad_pixels_info(self.handle, info, xy.ctypes.data, self.count)
        return xy[: self.count]

    def correct(self, frames):
        fr = _frames3(frames, torch.uint16)
        out = torch.empty_like(fr)
        _check(_lib.rir_bad_pixels_correct_device(self.handle, fr.data_ptr(), out.data_ptr(), fr.shape[0], _stream()), "rir_bad_pixels_correct_device")
        return out

    def remove_inplace(self, frames, rows):
        fr = _frames3(frames, torch.uint16)
        _check(_lib.rir_remove_bad_pixels_device(self.handle, fr.data_ptr(), int(rows), fr.shape[0], _stream()), "rir_remove_bad_pixels_device")
        return fr

    def close(self):
        if getattr(self, "handle", 0) > 0:
            _lib.bad_pixels_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def remove_motion(frames, shifts, rows=None):
    fr = _frames3(frames, torch.uint16)
    n, h, w = fr.shape
    sh = torch.as_tensor(shifts, dtype=torch.float32, device=fr.device).contiguous()
    out = torch.empty_like(fr)
    _check(_lib.rir_remove_motion_device(fr.data_ptr(), out.data_ptr(), w, h, h if rows is None else int(rows), n, sh.data_ptr(), _stream()),
           "rir_remove_motion_device")
    return out


def median_filter(frames):
    fr = _frames3(frames, torch.uint16)
    n, h, w = fr.shape
    out = torch.empty_like(fr)
    _check(_lib.rir_median_filter_device(fr.data_ptr(), out.data_ptr(), w, h, n, _stream()), "rir_median_filter_device")
    return out


def _label_args(image, background):
    if image.dim() != 2 or not image.is_cuda:
        raise RuntimeError("label_image: one (h, w) image on the device expected")
    img = image.contiguous()
    ch = _DTYPE_CHARS.get(img.dtype)
    if ch is None:
        raise RuntimeError("label_image: unsupported dtype")
    h, w = img.shape
    need = _lib.rir_label_workspace_bytes(w, h)
    if need == 0:
        raise RuntimeError("label_image: geometry refused")
    work = torch.empty(need // 8 + 1, dtype=torch.int64, device=img.device)
    back = np.zeros(1, dtype=_NP_OF[img.dtype])
    back[0] = background
    return img, ch, h, w, work, back


def label_image(image, background=0):
    """Connected components of one image in device memory (reference Filters.h:365-509): (labels int32 (h, w), areas, first-pixel table),
    everything on the device; entry 0 of the tables is the background's."""
    img, ch, h, w, work, back = _label_args(image, background)
    dst = torch.empty((h, w), dtype=torch.int32, device=img.device)
    xy = torch.empty((h * w + 1, 2), dtype=torch.float64, device=img.device)
    area = torch.empty(h * w + 1, dtype=torch.int32, device=img.device)
    count = torch.zeros(1, dtype=torch.int32, device=img.device)
    _check(_lib.rir_label_image_device(ord(ch), img.data_ptr(), dst.data_ptr(), w, h, back.ctypes.data, xy.data_ptr(), area.data_ptr(),
                                       count.data_ptr(), work.data_ptr(), work.numel() * 8, _stream()), "rir_label_image_device")
    r = int(count.item())
    return dst, area[:r], xy[:r]


def keep_largest_area(image, background=0, foreground=1):
    """`foreground` on the largest component, int(background) elsewhere (reference Filters.h:511-540), int32 (h, w) on the device."""
    img, ch, h, w, work, back = _label_args(image, background)
    dst = torch.empty((h, w), dtype=torch.int32, device=img.device)
    _check(_lib.rir_keep_largest_area_device(ord(ch), img.data_ptr(), dst.data_ptr(), w, h, back.ctypes.data, int(foreground), work.data_ptr(),
                                             work.numel() * 8, _stream()), "rir_keep_largest_area_device")
    return dst


def _label_batch_args(frames, background):
    fr = _frames3(frames)
    if not fr.is_cuda:
        raise RuntimeError("label_images: frames on the device expected")
    ch = _DTYPE_CHARS.get(fr.dtype)
    if ch is None:
        raise RuntimeError("label_images: unsupported dtype")
    n, h, w = fr.shape
    need = _lib.rir_label_workspace_bytes_batch(w, h, n)
    if need == 0:
        raise RuntimeError("label_images: geometry refused")
    work = torch.empty(need // 8 + 1, dtype=torch.int64, device=fr.device)
    back = np.zeros(1, dtype=_NP_OF[fr.dtype])
    back[0] = background
    return fr, ch, n, h, w, work, back


def label_images(frames, background=0, table_entries=None):
    """Connected components of every image of a batch (n, h, w) in device memory, five launches for the whole batch: (labels int32 (n, h, w),
    areas (n, table_entries), first-pixel table (n, table_entries, 2), counts (n,)), on the device.  counts[i] = components of image i + 1;
    a table with fewer entries than that holds the first ``table_entries`` of them (default: 1 024 entries, or h*w + 1 if that is less)."""
    fr, ch, n, h, w, work, back = _label_batch_args(frames, background)
    cap = min(1024, h * w + 1) if table_entries is None else int(table_entries)
    dst = torch.empty((n, h, w), dtype=torch.int32, device=fr.device)
    xy = torch.zeros((n, cap, 2), dtype=torch.float64, device=fr.device)
    area = torch.zeros((n, cap), dtype=torch.int32, device=fr.device)
    count = torch.zeros(n, dtype=torch.int32, device=fr.device)
    _check(_lib.rir_label_images_device(ord(ch), fr.data_ptr(), dst.data_ptr(), w, h, n, back.ctypes.data, xy.data_ptr(), area.data_ptr(), cap,
                                        count.data_ptr(), work.data_ptr(), work.numel() * 8, _stream()), "rir_label_images_device")
    return dst, area, xy, count


def keep_largest_areas(frames, background=0, foreground=1):
    """``foreground`` on the largest component of every image of a batch (n, h, w), int(background) elsewhere; int32 (n, h, w) on the device."""
    fr, ch, n, h, w, work, back = _label_batch_args(frames, background)
    dst = torch.empty((n, h, w), dtype=torch.int32, device=fr.device)
    _check(_lib.rir_keep_largest_areas_device(ord(ch), fr.data_ptr(), dst.data_ptr(), w, h, n, back.ctypes.data, int(foreground), work.data_ptr(),
                                              work.numel() * 8, _stream()), "rir_keep_largest_areas_device")
    return dst


def split_planes(frames, linesize=None, it=None):
    """uint16 frames -> (Y, U, V) byte planes [n][h][linesize] (reference h264.cpp:1066-1082)."""
    fr = _frames3(frames, torch.uint16)
    n, h, w = fr.shape
    ls = w if linesize is None else int(linesize)
    Y, U, V = (torch.zeros((n, h, ls), dtype=torch.uint8, device=fr.device) for _ in range(3))
    itp = None
    if it is not None:
        it = _frames3(it, torch.uint8)
        itp = it.data_ptr()
    _check(_lib.rir_split_planes_device(fr.data_ptr(), itp, w, h, n, ls, Y.data_ptr(), U.data_ptr(), V.data_ptr(), _stream()), "rir_split_planes_device")
    return Y, U, V


def merge_planes(Y, U, V, width, with_it=False):
    """(Y, U, V) byte planes -> uint16 frames (and the 8-bit image carried by Y) (reference h264.cpp:3016-3051)."""
    U = _frames3(U, torch.uint8)
    V = _frames3(V, torch.uint8)
    n, h, ls = U.shape
    img = torch.empty((n, h, width), dtype=torch.uint16, device=U.device)
    it = torch.empty((n, h, width), dtype=torch.uint8, device=U.device) if with_it else None
    Yp = _frames3(Y, torch.uint8).data_ptr() if Y is not None else None
    _check(_lib.rir_merge_planes_device(Yp, U.data_ptr(), V.data_ptr(), ls, width, h, n, img.data_ptr(), it.data_ptr() if with_it else None,
                                        _stream()), "rir_merge_planes_device")
    return (img, it) if with_it else img


class LossyStream:
    """Bounded-loss step on device-resident frames (reference H264_Saver::addImageLossyNoCamera / addLoss)."""

    def __init__(self, width, height, lossy_height=None, low_value_error=6, high_value_error=2, std_factor=5.0, running_average=32,
                 subtract_min=False, remove_bad_pixels=False):
        self.shape = (height, width)
        self.handle = _lib.rir_lossy_create(width, height, height if lossy_height is None else int(lossy_height), int(low_value_error),
                                            int(high_value_error), float(std_factor), int(running_average), int(bool(subtract_min)),
                                            int(bool(remove_bad_pixels)))
        if self.handle <= 0:
            raise RuntimeError("rir_lossy_create failed: %s" % last_error())

    def step(self, frames, add_loss=False, errors=True):
        """frames (n,h,w) uint16 on the device -> (processed frames, low_errors, high_errors).  With ``errors=False`` the
        frames are only queued on the current stream (nothing waits) and the two error arrays are None."""
        fr = _frames3(frames, torch.uint16)
        n = fr.shape[0]
        if tuple(fr.shape[1:]) != self.shape:
            raise RuntimeError("LossyStream.step: wrong frame size")
        out = torch.empty_like(fr)
        lo = np.zeros(n, np.int32) if errors else None
        hi = np.zeros(n, np.int32) if errors else None
        _check(_lib.rir_lossy_step_device(self.handle, fr.data_ptr(), out.data_ptr(), n, int(bool(add_loss)),
                                          lo.ctypes.data if errors else None, hi.ctypes.data if errors else None, _stream()),
               "rir_lossy_step_device")
        return out, lo, hi

    @staticmethod
    def step_many(streams, frames, add_loss=False, errors=True):
        """The step for several independent streams in shared launches (rir_lossy_step_multi_device): ``streams`` are
        LossyStream objects of one geometry that have seen the same number of frames, ``frames`` one (n,h,w) uint16 device
        tensor per stream.  -> (list of processed tensors, low_errors[stream][frame], high_errors[stream][frame])."""
        S = len(streams)
        frs = [_frames3(f, torch.uint16) for f in frames]
        if S == 0 or len(frs) != S or any(tuple(f.shape) != tuple(frs[0].shape) for f in frs) or tuple(frs[0].shape[1:]) != streams[0].shape:
            raise RuntimeError("LossyStream.step_many: one tensor of the streams' frame size per stream expected")
        n = frs[0].shape[0]
        outs = [torch.empty_like(f) for f in frs]
        handles = (ct.c_int * S)(*[s.handle for s in streams])
        pin = (ct.c_void_p * S)(*[f.data_ptr() for f in frs])
        pout = (ct.c_void_p * S)(*[o.data_ptr() for o in outs])
        lo = np.zeros((S, n), np.int32) if errors else None
        hi = np.zeros((S, n), np.int32) if errors else None
        _check(_lib.rir_lossy_step_multi_device(ct.cast(handles, _vp), S, ct.cast(pin, _vp), ct.cast(pout, _vp), n, int(bool(add_loss)),
                                                lo.ctypes.data if errors else None, hi.ctypes.data if errors else None, _stream()),
               "rir_lossy_step_multi_device")
        return outs, lo, hi

    def set_errors(self, low_value_error, high_value_error, std_factor):
        """lowValueError / highValueError / stdFactor from the next frame on (the budget's history stays)"""
        _check(_lib.rir_lossy_set_errors(self.handle, int(low_value_error), int(high_value_error), float(std_factor)), "rir_lossy_set_errors")

    def path_stats(self):
        """(groups of frames of the last batch this stream led that were offered to the constant-budget form, groups it took); waits"""
        out = (ct.c_int * 2)()
        _check(_lib.rir_lossy_path_stats(self.handle, out, _stream()), "rir_lossy_path_stats")
        return int(out[0]), int(out[1])

    def spec_stats(self):
        """the speculative form's books for the last batch this stream led: (groups through its launches, groups offered, groups committed,
        passes over all groups); waits"""
        out = (ct.c_int * 4)()
        _check(_lib.rir_lossy_spec_stats(self.handle, out, _stream()), "rir_lossy_spec_stats")
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def status(self):
        """raises when a queue-only ``step`` / ``step_many`` led by this stream went wrong on the device (waits for the stream)"""
        _check(_lib.rir_lossy_status(self.handle, _stream()), "rir_lossy_status")

    def close(self):
        if getattr(self, "handle", 0) > 0:
            _lib.rir_lossy_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
