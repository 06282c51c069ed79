"""``MaskedRegistratorECC`` for frames that already live in HBM (torch tensors on the device).

Same steps and arithmetic as ``masked_registration_ecc.MaskedRegistratorECC`` (reference
src/python/librir/registration/masked_registration_ecc.py:88-191) - gaussian pre-filter, centred crop, min-max
normalisation, translation-only ECC with the previous result as start value, change of reference image on a
confidence drop - with every pixel operation on the device: per frame one gaussian, two small normalisation
kernels, the ECC iterations and a read-back of (tx, ty, cc).  The static / percentile masks of the host class
are not offered here."""
import ctypes as ct

import numpy as np
import torch

from .. import device as D
from ..low_level.misc import _lib, last_error

_vp = ct.c_void_p
_lib.rir_minmax_normalize_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, _vp, _vp]
_lib.rir_ecc_translation_device.argtypes = [_vp, _vp, _vp, ct.c_int, ct.c_int, _vp, ct.c_int, ct.c_double, ct.POINTER(ct.c_double),
                                            ct.POINTER(ct.c_int), _vp]


_lib.rir_ecc_register_frame_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_float, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp,
                                               ct.c_int, ct.c_double, ct.POINTER(ct.c_double), ct.POINTER(ct.c_int), _vp]


_lib.rir_ecc_prepare_frames_device.argtypes = [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_float, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp,
                                               _vp]
_lib.rir_ecc_align_prepared_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_int, ct.c_int, _vp, ct.c_int, ct.c_double, ct.POINTER(ct.c_double),
                                               ct.POINTER(ct.c_int), _vp]


_lib.rir_ecc_align_prepared_frames_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, ct.c_int, ct.c_double, _vp, _vp]
_lib.rir_ecc_align_multi_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, ct.c_int, ct.c_double, _vp, ct.c_int, _vp, _vp]


class PrepareJob(ct.Structure):
    """rir_ecc_prepare_job (include/rir_amd_device.h): the arguments of one rir_ecc_prepare_frames_device call"""
    _fields_ = [("d_imgs", _vp), ("dtype", ct.c_int), ("w", ct.c_int), ("h", ct.c_int), ("nframes", ct.c_int), ("sigma", ct.c_float),
                ("win_x", ct.c_int), ("win_y", ct.c_int), ("win_w", ct.c_int), ("win_h", ct.c_int), ("d_norm", _vp), ("d_gx", _vp), ("d_gy", _vp)]


_lib.rir_ecc_align_multi_overlapped_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, ct.c_int, ct.c_double, _vp, ct.c_int, _vp,
                                                       ct.POINTER(PrepareJob), ct.c_int, _vp]


def _stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceRegistratorECC:
    def __init__(self, window_factorh=0.7, window_factorv=0.7, sigma=0.5, shape=(512, 640)):
        self.sigma = sigma
        self.x, self.y, self.confidences = [], [], []
        self.subW = int(shape[1] * window_factorh)
        self.subH = int(shape[0] * window_factorv)
        self.startX = int((shape[1] - self.subW) / 2)
        self.startY = int((shape[0] - self.subH) / 2)
        self.conf_thresh = None
        self.number_of_iterations = 500
        self.termination_eps = 1e-3
        self.warp = np.zeros(2, np.float32)
        self._ref_n = None

    def _filtered(self, img):
        img = img if img.dim() == 2 else img[0]
        if img.dtype != torch.float32 and img.dtype != torch.uint16:
            img = img.to(torch.float32)
        return D.gaussian_filter(img, self.sigma)[0] if self.sigma > 0 else img.to(torch.float32)

    def _normalised_window(self, full):
        """dense float32 (subH, subW): the min-max normalised registration window of a full filtered frame"""
        w = full.shape[1]
        win = full[self.startY:self.startY + self.subH, self.startX:self.startX + self.subW]  # strided view, not copied
        out = torch.empty((self.subH, self.subW), dtype=torch.float32, device=full.device)
        if _lib.rir_minmax_normalize_device(win.data_ptr(), self.subW, self.subH, w, out.data_ptr(), _stream()) != 0:
            raise RuntimeError("rir_minmax_normalize_device: %s" % last_error())
        return out

    def start(self, img):
        g = self._filtered(img)
        self._ref_full = g
        self._ref_win = None  # (the window of the current reference before normalisation: a view of _ref_full until the reference changes)
        self._ref_n = self._normalised_window(g)
        self.x.append(0)
        self.y.append(0)
        self.confidences.append(1)

    def reference_window(self):
        """the registration window of the current reference image, filtered, not normalised (what the host class calls ``ref_img``): device tensor"""
        if self._ref_win is not None:
            return self._ref_win
        return self._ref_full[self.startY:self.startY + self.subH, self.startX:self.startX + self.subW]

    def compute(self, img):
        img = img if img.dim() == 2 else img[0]
        if img.dtype != torch.float32 and img.dtype != torch.uint16:
            img = img.to(torch.float32)
        img = img.contiguous()
        cc = ct.c_double(0)
        # pre-filter, window normalisation and alignment in one library call (no host work between the kernels)
        if _lib.rir_ecc_register_frame_device(img.data_ptr(), ord("H") if img.dtype == torch.uint16 else ord("f"), img.shape[1], img.shape[0],
                                              float(self.sigma), self.startX, self.startY, self.subW, self.subH, self._ref_n.data_ptr(),
                                              self.warp.ctypes.data, self.number_of_iterations, self.termination_eps, ct.byref(cc), None,
                                              _stream()) != 0:
            raise RuntimeError("ECC: %s" % last_error())
        return self._after_alignment(img, cc.value)

    def _after_alignment(self, img, cc, frames=None, index=0):
        """book-keeping of compute(): records the shift; changes the reference image on a confidence drop (img, or frames[index])"""
        shift = [float(self.warp[1]), float(self.warp[0])]
        self.confidences.append(cc)
        self.x.append(shift[1])
        self.y.append(shift[0])
        if len(self.confidences) > 20:
            if self.conf_thresh is None:
                self.conf_thresh = np.min(self.confidences) - 2 * np.std(self.confidences)
            if cc < self.conf_thresh:
                self._change_reference(img, shift, frames, index)
        return shift

    def _change_reference(self, img, shift, frames=None, index=0):
        """change of reference image: the current window, shifted back (masked_registration_ecc.py:170-189)"""
        g = self._filtered(img if img is not None else frames[index])
        win = g[self.startY:self.startY + self.subH, self.startX:self.startX + self.subW].contiguous()
        moved = D.translate(win, (-shift[1], -shift[0]), "")[0]
        out = torch.empty_like(moved)
        if _lib.rir_minmax_normalize_device(moved.data_ptr(), self.subW, self.subH, self.subW, out.data_ptr(), _stream()) != 0:
            raise RuntimeError("rir_minmax_normalize_device: %s" % last_error())
        self._ref_n = out
        self._ref_win = moved
        self.warp[:] = 0

    def _as_frames(self, frames):
        if frames.dim() == 2:
            frames = frames[None]
        if frames.dtype != torch.float32 and frames.dtype != torch.uint16:
            frames = frames.to(torch.float32)
        return frames.contiguous()

    def _prepare(self, frames, c0, k, norm, st):
        n, h, w = frames.shape
        dt = ord("H") if frames.dtype == torch.uint16 else ord("f")
        if _lib.rir_ecc_prepare_frames_device(frames[c0].data_ptr(), dt, w, h, k, float(self.sigma), self.startX, self.startY, self.subW, self.subH,
                                              norm[0].data_ptr(), norm[1].data_ptr(), norm[2].data_ptr(), st) != 0:
            raise RuntimeError("rir_ecc_prepare_frames_device: %s" % last_error())

    def _prepare_job(self, frames, c0, k, norm):
        """what _prepare(frames, c0, k, norm, st) would do, as a job for rir_ecc_align_multi_overlapped_device"""
        n, h, w = frames.shape
        return PrepareJob(frames[c0].data_ptr(), ord("H") if frames.dtype == torch.uint16 else ord("f"), w, h, k, float(self.sigma), self.startX, self.startY,
                          self.subW, self.subH, norm[0].data_ptr(), norm[1].data_ptr(), norm[2].data_ptr())

    def _align(self, norm, i, cnt, res, st):
        good = _lib.rir_ecc_align_prepared_frames_device(self._ref_n.data_ptr(), norm[0, i].data_ptr(), norm[1, i].data_ptr(), norm[2, i].data_ptr(),
                                                         self.subW, self.subH, cnt, self.warp.ctypes.data, self.number_of_iterations, self.termination_eps,
                                                         res[i:].ctypes.data, st)
        if good < 0:
            raise RuntimeError("ECC: %s" % last_error())
        return good

    def _consume_chunk(self, frames, c0, k, norm, res, st, shifts, first_good=None, after_first=None):
        """The book-keeping of a chunk of k prepared frames: the alignments of the chunk in one launch (``first_good``: already
        done, by a multi-sequence launch, with ``res`` filled), then frame by frame what ``compute`` does; where that changes the
        reference image the rest of the chunk is aligned again (new reference, from the identity)."""
        i = 0
        while i < k:
            cnt = k - i
            if i == 0 and first_good is not None:
                good = first_good
            else:
                good = self._align(norm, i, cnt, res, st)
            if i == 0 and after_first is not None:
                after_first()
            # the frames up to the first one whose confidence falls below the threshold (it changes the reference image: what was
            # aligned after it does not count) are booked in one go - same values as frame-by-frame _after_alignment calls
            if good == cnt and self.conf_thresh is not None and (good == 0 or float(res[i:i + good, 2].min()) >= self.conf_thresh):
                # the ordinary round - every image aligned, no confidence below the (established) threshold - in four list operations
                blk = res[i:i + good]
                cols = blk.T.tolist()
                self.x.extend(cols[0])
                self.y.extend(cols[1])
                self.confidences.extend(cols[2])
                shifts.extend(blk[:, 1::-1].tolist())  # [y, x] pairs
                if good:
                    self.warp[0], self.warp[1] = blk[-1, 0], blk[-1, 1]
                return
            base = len(self.confidences)
            cc = res[i:i + good, 2]
            stop = good  # frames of this round that count
            change = False
            if good:
                if self.conf_thresh is None and base + good > 20:
                    first = max(0, 20 - base)  # the frame that makes it 21 confidences defines the threshold
                    head = np.array(self.confidences + cc[:first + 1].tolist())
                    self.conf_thresh = np.min(head) - 2 * np.std(head)
                    below = np.nonzero(cc[first:] < self.conf_thresh)[0]
                    if below.size:
                        stop, change = first + int(below[0]) + 1, True
                elif self.conf_thresh is not None:
                    below = np.nonzero(cc < self.conf_thresh)[0]
                    if below.size:
                        stop, change = int(below[0]) + 1, True
            xs, ys = res[i:i + stop, 0].tolist(), res[i:i + stop, 1].tolist()
            self.x.extend(xs)
            self.y.extend(ys)
            self.confidences.extend(cc[:stop].tolist())
            shifts.extend(res[i:i + stop, 1::-1].tolist())  # [y, x] pairs
            if stop:
                self.warp[0], self.warp[1] = res[i + stop - 1, 0], res[i + stop - 1, 1]
            if change:
                self._change_reference(None, [ys[-1], xs[-1]], frames, c0 + i + stop - 1)
            elif good < cnt:
                raise RuntimeError("ECC: the alignment did not converge (empty overlap, singular system or non-positive lambda) - %s" % last_error())
            i += stop

    def compute_many(self, frames, chunk=32):
        """``compute`` for every frame of a (n, h, w) device tensor, in order, with the same results.  The pre-processing of a
        chunk of frames (pre-filter, window normalisation, gradients) runs in shared launches, the alignments of the chunk in one
        launch, each from the previous shift; while the host does the book-keeping of a chunk the device prepares the next one.
        Returns the list of shifts."""
        frames = self._as_frames(frames)
        n = frames.shape[0]
        shifts = []
        m = min(chunk, n)
        bufs = [torch.empty((3, m, self.subH, self.subW), dtype=torch.float32, device=frames.device) for _ in range(2 if n > chunk else 1)]
        st = _stream()
        if n:
            self._prepare(frames, 0, min(chunk, n), bufs[0], st)
        for ci, c0 in enumerate(range(0, n, chunk)):
            k = min(chunk, n - c0)
            norm = bufs[ci % len(bufs)]
            res = np.empty((k, 4), np.float64)
            nxt = None
            if c0 + chunk < n:  # the next chunk's pre-processing is queued before the book-keeping of this one
                nxt = lambda c0=c0, ci=ci: self._prepare(frames, c0 + chunk, min(chunk, n - c0 - chunk), bufs[(ci + 1) % len(bufs)], st)  # noqa: E731
            self._consume_chunk(frames, c0, k, norm, res, st, shifts, after_first=nxt)
        return shifts

    @staticmethod
    def compute_many_multi(registrators, frames, chunk=64):
        """``compute_many`` for S independent sequences at once - ``registrators[q]`` (each started on its own reference image, all
        with one window size) tracks ``frames[q]`` (n, h, w) - with the alignments of a chunk of ALL sequences in one resident
        launch (rir_ecc_align_multi_device): an alignment is a chain of dependent iterations that cannot fill the chip, S chains
        side by side can.  Every sequence gets exactly the track its own ``compute_many`` gives.  Returns the lists of shifts."""
        S = len(registrators)
        if S == 0 or len(frames) != S:
            raise RuntimeError("compute_many_multi: one frames tensor per registrator expected")
        frs = [r._as_frames(f) for r, f in zip(registrators, frames)]
        n = frs[0].shape[0]
        r0 = registrators[0]
        if any(f.shape[0] != n for f in frs) or any((r.subW, r.subH, r.number_of_iterations, r.termination_eps) !=
                                                    (r0.subW, r0.subH, r0.number_of_iterations, r0.termination_eps) for r in registrators):
            raise RuntimeError("compute_many_multi: the sequences must share length, window size and termination criteria")
        st = _stream()
        # The chunks: a SHORT first one - its pre-processing is the only one that nothing hides (every later chunk's runs under the
        # alignments of the chunk before) - then chunks of equal size, none larger than asked for (a last chunk of three images
        # costs a launch like any other).
        first = min(n, max(4, chunk // 4)) if n > chunk else n
        rest = n - first
        if rest > chunk:
            chunk = -(-rest // -(-rest // chunk))
        starts = [0] + list(range(first, n, chunk))
        sizes = [first] + [min(chunk, n - c) for c in starts[1:]]
        m = max(sizes)
        dev = frs[0].device
        bufs = [[torch.empty((3, m, r0.subH, r0.subW), dtype=torch.float32, device=dev) for _ in range(S)] for _ in range(2 if len(starts) > 1 else 1)]
        shifts = [[] for _ in range(S)]
        ptr = lambda ts: (ct.c_void_p * S)(*[t.data_ptr() for t in ts])  # noqa: E731
        for q in range(S):
            if n:
                registrators[q]._prepare(frs[q], 0, first, bufs[0][q], st)
        # The pre-processing of chunk k + 1 runs UNDER the alignments of chunk k - but it is the library that starts it
        # (rir_ecc_align_multi_overlapped_device), and only once the alignment launch has reported itself resident: that launch
        # needs every one of its workgroups on the chip to start, and ordinary kernels that come and go beside it before that
        # leave the register files fragmented and keep the last ones out (the residency gate: queued from here on
        # a second stream it failed 10 launches of 12).
        for ci, (c0, k) in enumerate(zip(starts, sizes)):
            if k == 0:
                break
            norm = bufs[ci % len(bufs)]
            res = np.empty((S, k, 4), np.float64)
            warps = np.stack([r.warp for r in registrators]).astype(np.float32)
            counts = (ct.c_int * S)(*([k] * S))
            good = (ct.c_int * S)()
            jobs = (PrepareJob * S)()
            njobs = 0
            if ci + 1 < len(starts):  # the next chunk's pre-processing: beside the alignments of this one
                njobs = S
                for q in range(S):
                    jobs[q] = registrators[q]._prepare_job(frs[q], starts[ci + 1], sizes[ci + 1], bufs[(ci + 1) % len(bufs)][q])
            if _lib.rir_ecc_align_multi_overlapped_device(ptr([r._ref_n for r in registrators]), ptr([b[0] for b in norm]), ptr([b[1] for b in norm]),
                                                          ptr([b[2] for b in norm]), r0.subW, r0.subH, S, counts, warps.ctypes.data,
                                                          r0.number_of_iterations, r0.termination_eps, res.ctypes.data, k, good, jobs, njobs, st) != 0:
                raise RuntimeError("ECC: %s" % last_error())
            for q in range(S):
                registrators[q]._consume_chunk(frs[q], c0, k, norm[q], res[q], st, shifts[q], first_good=int(good[q]))
        return shifts

    def return_coordinates_and_confidence_values(self):
        return np.array([self.x, self.y, self.confidences]).T

    def to_reg_file(self, dest_file):
        arr = self.return_coordinates_and_confidence_values()
        with open(dest_file, "w") as f:
            f.write("\tx-axis translations\ty-axis translations\tConfidence level\n")
            for i, (x, y, c) in enumerate(arr):
                f.write("%d\t%s\t%s\t%s\n" % (i, repr(float(x)), repr(float(y)), repr(float(c))))
