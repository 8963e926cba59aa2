"""Host-side handle on a libttscale plan: the Python mirror of "give ffmpeg a -vf string".

The reference drives this path through filter strings only
(scripts/processVideos.js:94-97, src/services/video.js:329-332); `Plan.from_vf` takes the
same strings.  Frames are packed I420 (`-f rawvideo -pix_fmt yuv420p`) uint8 tensors that
already live in HBM; torch is used for device memory and streams, nothing else."""
import ctypes

import numpy as np

from . import _lib
from ._lib import Geom, Frame, TtsError, check

ALG = {"bicubic": 0, "area": 1, "bilinear": 2}


def frame_bytes(w, h):
    return int(_lib.lib().tts_frame_bytes(w, h))


FMT = {"yuv420p": 0, "rgb24": 1, "bgr24": 2, "rgba": 3, "bgra": 4, "yuvj420p": 5}   # (yuvj420p: as an INPUT format only -- full-range planes)


def parse_vf(vf, iw, ih, in_fmt="yuv420p"):
    """Resolve a filter chain for an iw x ih input -> Geom (raises TtsError).  in_fmt: "yuv420p", or "rgb24" /
    "bgr24" / "rgba" / "bgra" for a chain fed with packed RGB (`-pix_fmt rgb24 -i ...`), or "yuvj420p" for yuv420p planes that
    hold full-range samples (the geometry then says what differs: sc_range / out_range, the pad colour)."""
    g = Geom()
    rc = _lib.lib().tts_parse_vf_fmt(vf.encode(), iw, ih, FMT[in_fmt], ctypes.byref(g))
    check(rc, _lib.lib().tts_last_error().decode() if rc else "")
    return g


def build_filter(src, dst, one, alg):
    """Host-only integer resampling table: (taps[dst, size] int16, pos[dst] int32)."""
    taps = ctypes.c_int()
    check(_lib.lib().tts_build_filter(src, dst, one, alg, ctypes.byref(taps), None, None))
    coef = np.zeros((dst, taps.value), np.int16)
    pos = np.zeros(dst, np.int32)
    check(_lib.lib().tts_build_filter(src, dst, one, alg, ctypes.byref(taps),
                                      coef.ctypes.data_as(ctypes.c_void_p), pos.ctypes.data_as(ctypes.c_void_p)))
    return coef, pos


def _fps(fps):
    from fractions import Fraction
    f = Fraction(fps).limit_denominator(1001 * 64)
    return f.numerator, f.denominator


def trim_range(n_frames, fps, start, duration=None):
    """(first, count) of `-ss start -t duration` on an n_frames clip (scripts/processVideos.js:84-92)."""
    num, den = _fps(fps)
    first, count = ctypes.c_int(), ctypes.c_int()
    check(_lib.lib().tts_trim_range(n_frames, num, den, float(start), -1.0 if duration is None else float(duration),
                                    ctypes.byref(first), ctypes.byref(count)))
    return first.value, count.value


def loop_order(n_frames, fps, extract_start, extract_duration, repeats):
    """Source frame of every output frame of createLoopedVideo (scripts/loopVideos.js:54-100)."""
    num, den = _fps(fps)
    n = _lib.lib().tts_loop_order(n_frames, num, den, float(extract_start), float(extract_duration), repeats, None, 0)
    if n < 0:
        check(n)
    order = np.zeros(n, np.int32)
    got = _lib.lib().tts_loop_order(n_frames, num, den, float(extract_start), float(extract_duration), repeats,
                                    order.ctypes.data_as(ctypes.c_void_p), n)
    if got < 0:
        check(got)
    return order


class Plan:
    """Immutable per-geometry state (coefficient tables in HBM on the current device)."""

    def __init__(self, geom):
        self._h = ctypes.c_void_p()
        self.geom = geom
        rc = _lib.lib().tts_plan_create(ctypes.byref(geom), ctypes.byref(self._h))
        check(rc, _lib.lib().tts_last_error().decode() if rc else "")

    @classmethod
    def from_vf(cls, vf, iw, ih, in_fmt="yuv420p"):
        return cls(parse_vf(vf, iw, ih, in_fmt))

    def close(self):
        if self._h:
            _lib.lib().tts_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def in_size(self):
        return self.geom.in_w, self.geom.in_h

    @property
    def out_size(self):
        return self.geom.out_w, self.geom.out_h

    @property
    def in_bytes(self):
        return int(_lib.lib().tts_plan_frame_bytes(self._h, 0))

    @property
    def out_bytes(self):
        """Bytes of one output frame: packed I420, or out_w * out_h * bpp when the chain ends in packed RGB."""
        return int(_lib.lib().tts_plan_frame_bytes(self._h, 1))

    @property
    def out_bpp(self):
        """Bytes per pixel of a packed-RGB output (0 for planar yuv420p)."""
        return {0: 0, 1: 3, 2: 3, 3: 4, 4: 4}[self.geom.out_fmt]

    def kernel_kind(self, plane_class):
        return _lib.lib().tts_plan_kernel_kind(self._h, plane_class)

    def kernel_name(self, plane_class):
        return _lib.lib().tts_plan_kernel_name(self._h, plane_class).decode()

    def filter(self, plane_class, axis):
        """(taps[n_out, size] int16, pos[n_out] int32) of one plane class / axis."""
        taps, n = ctypes.c_int(), ctypes.c_int()
        check(_lib.lib().tts_plan_filter(self._h, plane_class, axis, ctypes.byref(taps), ctypes.byref(n), None, None))
        coef = np.zeros((n.value, taps.value), np.int16)
        pos = np.zeros(n.value, np.int32)
        check(_lib.lib().tts_plan_filter(self._h, plane_class, axis, None, None,
                                         coef.ctypes.data_as(ctypes.c_void_p), pos.ctypes.data_as(ctypes.c_void_p)))
        return coef, pos

    @property
    def device(self):
        """Index of the GPU the plan's tables live on (the device that was current when it was made)."""
        return _lib.lib().tts_plan_device(self._h)

    def _check_tensor(self, t, what):
        """A frame buffer must be a contiguous uint8 tensor on the plan's own GPU: the kernels are launched
        on that tensor's stream with the plan's tables, so a tensor on another device would mean a launch
        with foreign pointers instead of a clean INVALID_INPUT."""
        import torch
        if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
            raise TtsError(-1, "%s must be a contiguous uint8 CUDA tensor" % what)
        if t.device.index != self.device:
            raise TtsError(-1, "%s is on cuda:%d, the plan was made on cuda:%d" % (what, t.device.index, self.device))

    def run_batch(self, src, dst=None, stream=None):
        """src: uint8 cuda tensor [n, in_bytes] (or flat); returns dst [n, out_bytes].
        Asynchronous on `stream` (default: torch's current stream)."""
        import torch
        self._check_tensor(src, "src")
        if src.numel() % self.in_bytes:
            raise TtsError(-1, "src size is not a multiple of the packed frame size")
        n = src.numel() // self.in_bytes
        if dst is None:
            dst = torch.empty((n, self.out_bytes), dtype=torch.uint8, device=src.device)
        self._check_tensor(dst, "dst")
        if dst.numel() != n * self.out_bytes:
            raise TtsError(-1, "dst must hold n*out_bytes bytes")
        if stream is None:
            stream = torch.cuda.current_stream(src.device).cuda_stream
        check(_lib.lib().tts_run_batch(self._h, src.data_ptr(), self.in_bytes, dst.data_ptr(), self.out_bytes, n,
                                       ctypes.c_void_p(stream)))
        return dst

    def strided_bytes(self, which, stride_y, stride_c):
        """Bytes of one frame (which = "in" / "out") whose rows lie stride_y / stride_c bytes apart."""
        w, h = self.in_size if which == "in" else self.out_size
        if which != "in" and self.geom.out_fmt:
            return stride_y * h                      # packed RGB: one plane
        return stride_y * h + 2 * stride_c * ((h + 1) // 2)

    def run_batch_strided(self, src, src_strides, dst, dst_strides, n, stream=None):
        """Packed batches with padded rows: src / dst are uint8 CUDA tensors holding n frames of
        strided_bytes(...) each; *_strides = (luma stride, chroma stride)."""
        import torch
        sp, dp = self.strided_bytes("in", *src_strides), self.strided_bytes("out", *dst_strides)
        self._check_tensor(src, "src")
        self._check_tensor(dst, "dst")
        if src.numel() < n * sp or dst.numel() < n * dp:
            raise TtsError(-1, "buffers are smaller than n strided frames")
        if stream is None:
            stream = torch.cuda.current_stream(src.device).cuda_stream
        check(_lib.lib().tts_run_batch_strided(self._h, src.data_ptr(), sp, src_strides[0], src_strides[1], dst.data_ptr(), dp,
                                               dst_strides[0], dst_strides[1], n, ctypes.c_void_p(stream)))
        return dst

    def run_batch_order(self, src, order, dst=None, stream=None):
        """Like run_batch, but output frame j comes from source frame order[j] (trimmed / looped clips)."""
        import torch
        self._check_tensor(src, "src")
        if src.numel() % self.in_bytes:
            raise TtsError(-1, "src must hold whole frames")
        order = np.ascontiguousarray(order, dtype=np.int32)
        n_src, n = src.numel() // self.in_bytes, int(order.size)
        if dst is None:
            dst = torch.empty((n, self.out_bytes), dtype=torch.uint8, device=src.device)
        self._check_tensor(dst, "dst")
        if dst.numel() != n * self.out_bytes:
            raise TtsError(-1, "dst must hold n*out_bytes bytes")
        if stream is None:
            stream = torch.cuda.current_stream(src.device).cuda_stream
        check(_lib.lib().tts_run_batch_order(self._h, src.data_ptr(), self.in_bytes, n_src, dst.data_ptr(), self.out_bytes,
                                             order.ctypes.data_as(ctypes.c_void_p), n, ctypes.c_void_p(stream)))
        return dst

    def run_frames(self, src_planes, dst_planes, src_strides, dst_strides, stream=None, host=False):
        """Arbitrary plane pointers: lists (one entry per frame) of 3 pointers / 3 strides (packed RGB out:
        plane 0 / stride 0 alone).  host=True: the pointers are host memory (TTS_MEM_HOST; synchronous)."""
        n = len(src_planes)
        S, D = (Frame * n)(), (Frame * n)()
        for i in range(n):
            for k in range(3):
                S[i].plane[k] = src_planes[i][k]
                D[i].plane[k] = dst_planes[i][k]
                S[i].stride[k] = src_strides[i][k]
                D[i].stride[k] = dst_strides[i][k]
            S[i].width, S[i].height = self.in_size
            D[i].width, D[i].height = self.out_size
            S[i].mem = D[i].mem = 1 if host else 0
        if host:
            check(_lib.lib().tts_run(self._h, S, D, n, None), _lib.lib().tts_last_error().decode())
            return
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.lib().tts_run(self._h, S, D, n, ctypes.c_void_p(stream)))

    def run_host(self, src, out=None):
        """numpy uint8 [n, in_bytes] on the host -> numpy [n, out_bytes] (`out`, if given: a C-contiguous
        uint8 array of that size, e.g. a view of pinned memory); pipelined through HBM in chunks."""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if src.size % self.in_bytes:
            raise TtsError(-1, "src holds %d bytes: not a whole number of %d-byte frames" % (src.size, self.in_bytes))
        n = src.size // self.in_bytes
        if out is not None and not (out.dtype == np.uint8 and out.flags.c_contiguous and out.size == n * self.out_bytes):
            raise TtsError(-1, "out must be a C-contiguous uint8 array of n*out_bytes")
        dst = np.empty((n, self.out_bytes), np.uint8) if out is None else out
        check(_lib.lib().tts_run_host(self._h, src.ctypes.data_as(ctypes.c_void_p),
                                      dst.ctypes.data_as(ctypes.c_void_p), n))
        return dst

    def run_host_static(self, src, out, stream=None):
        """tts_run_host_static: like run_host, staged through the library's static device scratch (no
        allocation) on `stream`, NOT synchronised: `out` (required; pinned memory makes the copies
        asynchronous) is complete once the stream has passed the call."""
        # (the call returns before its copies have run: a converted temporary of `src` would be freed under them)
        if not (isinstance(src, np.ndarray) and src.dtype == np.uint8 and src.flags.c_contiguous):
            raise TtsError(-1, "src must be a C-contiguous uint8 array (it is read after the call returns)")
        if src.size % self.in_bytes:
            raise TtsError(-1, "src holds %d bytes: not a whole number of %d-byte frames" % (src.size, self.in_bytes))
        n = src.size // self.in_bytes
        if not (out.dtype == np.uint8 and out.flags.c_contiguous and out.size == n * self.out_bytes):
            raise TtsError(-1, "out must be a C-contiguous uint8 array of n*out_bytes")
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = _lib.lib().tts_run_host_static(self._h, src.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                                            n, ctypes.c_void_p(stream))
        check(rc, _lib.lib().tts_last_error().decode() if rc else "")
        return out
